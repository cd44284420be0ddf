"""Diagnostic: a multi-group, pipelined ctx against the single-group lock-step ctx on the same
sequences, repeated; on a mismatch prints which sequence diverged first, at which frame, and in what.
Usage: group_stress.py [--config tiny] [--seqs 12] [--frames 14] [--groups 2,3] [--repeats 5]
                       [--motion 4.0] [--concurrent]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "stereo-svo-slam_amd")]
import numpy as np
import torch
from stereo_svo_slam_amd import synth
from stereo_svo_slam_amd.stereo_slam import StereoSlamBatch

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="tiny")
ap.add_argument("--seqs", type=int, default=12)
ap.add_argument("--frames", type=int, default=14)
ap.add_argument("--groups", default="2,3")
ap.add_argument("--repeats", type=int, default=5)
ap.add_argument("--motion", type=float, default=4.0)
ap.add_argument("--concurrent", action="store_true", help="run the lock-step ctx between the submits (as the test does)")
ap.add_argument("--borrow", action="store_true")
a = ap.parse_args()

if a.config == "tiny":
    seqs = [synth.make_sequence(a.config, a.frames, s, device="cpu", motion_scale=a.motion) for s in range(a.seqs)]
else:      # fused GPU renderer (full-size frames), host copies for the lock-step ctx
    seqs = []
    for s in range(a.seqs):
        cfg_, L, R, poses, ts = synth.make_sequence_gpu(a.config, a.frames, s, motion_scale=a.motion)
        seqs.append((cfg_, [x.cpu() for x in L], [x.cpu() for x in R], poses, ts))
cfg = seqs[0][0]
dev_l = [[s[1][k].cuda() for s in seqs] for k in range(a.frames)]
dev_r = [[s[2][k].cuda() for s in seqs] for k in range(a.frames)]
torch.cuda.synchronize()


def snapshot(ctx):
    out = []
    for i in range(a.seqs):
        f = ctx.get_frame(i)
        out.append(dict(pose=f.pose, k2=f.kps2d, k3=f.kps3d, info=f.info, traj=ctx.get_trajectory(i),
                        nkf=ctx.num_keyframes(i)))
    return out


def lockstep():
    os.environ["SVO_GROUPS"] = "1"
    one = StereoSlamBatch(cfg, cfg["width"], cfg["height"], a.seqs)
    return one


def feed_one(one, k):
    one.new_images([s[1][k].numpy() for s in seqs], [s[2][k].numpy() for s in seqs], [float(s[4][k]) for s in seqs])


ref_ctx = lockstep()
for k in range(a.frames):
    feed_one(ref_ctx, k)
ref = snapshot(ref_ctx)
ref_ctx.close()
print("reference: keyframes per sequence", [r["nkf"] for r in ref], "keypoints", [len(r["k2"]) for r in ref], flush=True)

bad = 0
for g in [int(x) for x in a.groups.split(",")]:
    for rep in range(a.repeats):
        os.environ["SVO_GROUPS"] = str(g)
        multi = StereoSlamBatch(cfg, cfg["width"], cfg["height"], a.seqs)
        assert multi.groups() == g
        packs = [multi.pack_images(dev_l[k], dev_r[k], [float(s[4][k]) for s in seqs], borrow=a.borrow) for k in range(a.frames)]
        one = lockstep() if a.concurrent else None
        for k in range(a.frames):
            if one is not None:
                feed_one(one, k)
            multi.submit_packed(packs[k])
        multi.wait()
        got = snapshot(multi)
        if one is not None:
            got1 = snapshot(one)
            one.close()
        else:
            got1 = None
        multi.close()
        for name, res in (("multi", got), ("lockstep-concurrent", got1)):
            if res is None:
                continue
            for i in range(a.seqs):
                r, q = ref[i], res[i]
                same = (np.array_equal(r["pose"], q["pose"]) and np.array_equal(r["k2"], q["k2"]) and
                        np.array_equal(r["k3"], q["k3"]) and np.array_equal(r["info"], q["info"]) and
                        np.array_equal(r["traj"], q["traj"]) and r["nkf"] == q["nkf"])
                if same:
                    continue
                bad += 1
                n = min(len(r["traj"]), len(q["traj"]))
                d = np.any(r["traj"][:n] != q["traj"][:n], axis=1)
                first = int(np.argmax(d)) if d.any() else -1
                print(f"MISMATCH groups={g} rep={rep} ctx={name} seq={i}: first differing trajectory frame {first}, "
                      f"keypoints {len(r['k2'])} vs {len(q['k2'])}, keyframes {r['nkf']} vs {q['nkf']}", flush=True)
                if first >= 0:
                    print("   ref ", r["traj"][first], "\n   got ", q["traj"][first], flush=True)
        print(f"groups={g} rep={rep} done, mismatches so far {bad}", flush=True)
print("TOTAL MISMATCHES", bad)
sys.exit(1 if bad else 0)
