"""Diagnostic: B sequences as G independent contexts (own stream + host thread each), free-running.
Shows how much of the step time is host round trips / single-stream serialisation.
Usage: groups_experiment.py [B] [steps] [G ...]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stereo-svo-slam_amd"))
import torch
import bench
from stereo_svo_slam_amd.stereo_slam import StereoSlamBatch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
Gs = [int(g) for g in sys.argv[3:]] or [1, 2, 4]
W = 4
dev = torch.device("cuda", 0)
cfg, lefts, rights, ts = bench.render_sequences("euroc", list(range(B)), W + K, dev)
torch.cuda.synchronize()
for G in Gs:
    n = B // G
    slams, packs = [], []
    for g in range(G):
        s = StereoSlamBatch(cfg, cfg["width"], cfg["height"], n, 0)
        if os.environ.get("TIMING"):
            s.enable_timing(True)
        slams.append(s)
        packs.append([s.pack_images([lefts[g * n + i][k] for i in range(n)], [rights[g * n + i][k] for i in range(n)],
                                    [float(ts[k])] * n) for k in range(W + K)])
    def run(g, k0, k1):
        for k in range(k0, k1):
            slams[g].new_images_packed(packs[g][k])
    def phase(k0, k1):
        th = [threading.Thread(target=run, args=(g, k0, k1)) for g in range(G)]
        t = time.perf_counter()
        for x in th: x.start()
        for x in th: x.join()
        torch.cuda.synchronize()
        return time.perf_counter() - t
    phase(0, W)
    dt = phase(W, W + K)
    print(f"G={G}: {1e3 * dt / K:.3f} ms/step  {B * K / dt:.0f} frames/s", flush=True)
    for s in slams: s.close()
