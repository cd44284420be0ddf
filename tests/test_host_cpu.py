"""CPU tests of the host side: the C-ABI library loads and exports exactly what
include/svo_hip.h declares (no compute without a GPU), it refuses to run
without a device instead of falling back, and the multi-GPU driver logic
works over gloo with world_size 2."""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from stereo_svo_slam_amd import hip_lib, multi_seq, synth
import util


def _free_port():
    """a TCP port nobody listens on right now (the rendezvous of the two-rank tests)"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "svo_hip.h")).read()
    declared = set(re.findall(r"\b(svo_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(hip_lib.SYMBOLS), declared ^ set(hip_lib.SYMBOLS)
    lib = hip_lib.lib()
    for sym in declared:
        assert hasattr(lib, sym), sym
    lib.svo_version.restype = C.c_int
    assert lib.svo_version() >= 100


def test_struct_layouts_match_the_header():
    assert C.sizeof(hip_lib.CameraSettings) == 10 * 4 + 9 * 4
    from stereo_svo_slam_amd import stereo_slam
    assert stereo_slam.KP_INFO_DTYPE.itemsize == 44
    assert C.sizeof(stereo_slam.GnTrace) == 52
    assert C.sizeof(stereo_slam.FrameStats) == 6 * 4 + 12 * 4 + 3 * 4 + 8 * 4 + 9 * 52
    assert C.sizeof(stereo_slam.Totals) == 5 * 8 + 8 * 8 + 8 + 8 + 4 + 4


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback_without_gpu():
    """The product path must fail loudly when no GPU is there."""
    lib = hip_lib.lib()
    h = C.c_void_p()
    rc = lib.svo_handle_create(0, 128, C.byref(h))
    assert rc < 0 and b"no HIP device" in lib.svo_last_error()
    cam = hip_lib.CameraSettings.from_dict(synth.CONFIGS["tiny"])
    ctx = C.c_void_p()
    rc = lib.svo_ctx_create(C.byref(cam), 320, 240, 1, 0, C.byref(ctx))
    assert rc < 0
    with pytest.raises(hip_lib.SvoError):
        hip_lib.Handle(0)
    from stereo_svo_slam_amd.stereo_slam import StereoSlamBatch
    with pytest.raises(hip_lib.SvoError):
        StereoSlamBatch(synth.CONFIGS["tiny"], 320, 240, 1)


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "stereo-svo-slam_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_py" not in txt and "svo_oracle" not in txt and "libsvo_oracle" not in txt, f


def test_sequence_sharding_is_a_partition():
    ids = sum((multi_seq.sequence_ids(r, 4, 3) for r in range(4)), [])
    assert sorted(ids) == list(range(12))


def test_longest_first_assignment_of_c4_sequences():
    """C4: six sequences of unequal length on N ranks, longest first, every sequence
    on exactly one rank; with eight ranks six stay busy, with two the loads differ by < 10 %."""
    lengths = [2912, 1710, 2149, 2280, 2348, 1922]          # the EuRoC V1_01..V2_03 image counts of SURVEY §8d
    for n in (1, 2, 4, 8):
        parts = multi_seq.assign_longest_first(lengths, n)
        assert len(parts) == n and sorted(sum(parts, [])) == list(range(6))
        loads = [sum(lengths[i] for i in p) for p in parts]
        assert max(loads) >= max(lengths)
        if n == 8:
            assert sum(1 for p in parts if p) == 6
        if n == 2:
            assert (max(loads) - min(loads)) / sum(loads) < 0.1
    assert multi_seq.assign_longest_first(lengths, 1)[0] == [0, 4, 3, 2, 5, 1]


def test_synthetic_stereo_geometry():
    """`right` is displaced toward -x: a point at depth z shifts by +baseline/z columns."""
    cfg, L, R, poses, ts = synth.make_sequence("tiny", 1, 0, device="cpu", noise_sigma=0.0)
    import oracle_py as O
    kps = np.float32([[160, 120], [100, 80], [220, 150]])
    d = O.ssd_disparity(L[0].numpy(), R[0].numpy(), kps, 21, 30, 4, 1)
    assert np.all(d > 2) and np.all(d < 30)      # depths 0.7 m .. 10 m with baseline*fx = 20


WORKER = r"""
import json, os, sys
sys.path[:0] = [os.path.join(ROOT, "stereo-svo-slam_amd"), os.path.join(ROOT, "oracle")]
import numpy as np, torch
from stereo_svo_slam_amd import multi_seq, synth
import oracle_py as O
rank, local_rank, world = multi_seq.init_distributed("gloo")
ids = multi_seq.sequence_ids(rank, world, 2)
seqs = [synth.make_sequence("tiny", 4, s, device="cpu") for s in ids]
cam = O.make_camera(**{k: seqs[0][0][k] for k in synth.CAMERA_FIELDS})
slams = [O.Slam(cam) for _ in ids]
def step(k):
    for s, sl in zip(seqs, slams):
        sl.new_image(s[1][k].numpy(), s[2][k].numpy(), float(s[4][k]))
sec = multi_seq.timed_steps(step, 3, 1, world, None)
local = [[sid, 4] + [float(v) for v in sl.pose()] for sid, sl in zip(ids, slams)]
allsum = multi_seq.gather_summaries(local, world, None)
if rank == 0:
    print(json.dumps({"seconds": sec, "summaries": allsum.tolist(), "fps": multi_seq.throughput(2 * 3 * world, sec)}))
"""


def test_two_rank_gloo_run(tmp_path):
    """world_size 2 over gloo: barrier/max timing, sharding by sequence, one all_gather."""
    script = tmp_path / "worker.py"
    script.write_text(f"ROOT = {ROOT!r}\n" + WORKER)
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                          "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
                          port, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    summ = np.array(res["summaries"])
    assert summ.shape == (4, 8) and sorted(summ[:, 0].tolist()) == [0, 1, 2, 3]
    assert res["seconds"] > 0 and res["fps"] > 0
    # every rank's sequences give the same result as a single-process run
    import oracle_py as O
    for row in summ:
        cfg, L, R, poses, ts = synth.make_sequence("tiny", 4, int(row[0]), device="cpu")
        s = O.Slam(util.oracle_camera(cfg))
        for k in range(4):
            s.new_image(L[k].numpy(), R[k].numpy(), float(ts[k]))
        assert np.allclose(row[2:], s.pose(), atol=0)


def test_cpp_facade_builds_and_fails_loudly_without_gpu(tmp_path):
    """The C++ facade (reference class and method names, hostcpp/stereo_slam.hpp) compiles and links
    against libsvo_hip.so; without a GPU StereoSlam::new_image throws the library's error text
    instead of computing anything on the CPU."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "stereo-svo-slam_amd", "csrc")
    exe = str(tmp_path / "facade_smoke")
    gxx = shutil.which("g++")
    assert gxx
    subprocess.check_call([gxx, "-std=c++17", "-O1", os.path.join(root, "tests", "cpp", "facade_smoke.cpp"),
                           "-o", exe, "-L" + csrc, "-lsvo_hip", "-Wl,-rpath," + csrc,
                           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    # the stage classes (PoseEstimator, PoseRefiner, OpticalFlow, DepthFilter) build with g++ alone too
    subprocess.check_call([gxx, "-std=c++17", "-O1", os.path.join(root, "tests", "cpp", "facade_stages.cpp"),
                           "-o", str(tmp_path / "facade_stages"), "-L" + csrc, "-lsvo_hip", "-Wl,-rpath," + csrc,
                           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    import torch
    if torch.cuda.is_available():
        assert r.returncode == 0 and "keypoints" in r.stdout, (r.returncode, r.stdout, r.stderr)
    else:
        assert r.returncode == 3 and "no HIP device" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_timed_steps_runs_finish_fn_inside_the_timed_region():
    """Queued steps (svo_submit_images) are drained by finish_fn before the clock stops: once after
    the warm-up steps and once after the timed ones."""
    calls = []
    dt = multi_seq.timed_steps(lambda k: calls.append(("step", k)), 3, 2, 1, None,
                               finish_fn=lambda: calls.append(("finish",)))
    assert dt >= 0
    assert calls == [("step", 0), ("step", 1), ("finish",), ("step", 2), ("step", 3), ("step", 4), ("finish",)]


def test_bench_algorithmic_bytes_follow_design_section_5():
    """bench.algorithmic_bytes restates DESIGN.md section 5 / SURVEY 8d for the C2 configuration."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    cfg = synth.CONFIGS["euroc"]
    n = 150
    ab = bench.algorithmic_bytes(cfg, n, n)
    W, H = 752, 480
    b_p = 2 * W * H + sum((W >> l) * (H >> l) for l in range(1, 6)) + 376 * 240 + 188 * 120
    assert ab["images+pyramids"] == b_p
    assert ab["sparse_align"] == 4 * n * 120
    assert ab["klt"] == n * 3 * 2 * 33 * 33 + 21 * n
    assert ab["ssd_disparity"] == n * (31 * 31 + (31 + 60) * (31 + 12)) + 12 * n
    assert ab["reproj_gn"] == 24 * n and ab["filter_update"] == 64 * n


def test_bench_loop_plan_and_group_sample():
    """bench.py's workload plan: ctx sequence s plays loop s % n_loops from its own entry frame, always
    forward; sequences sharing a loop never show the same frame at the same step; the parity sample
    takes two sequences of every sequence group, the first G entries covering all groups."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    plan = bench.loop_plan(2048, 128, 192)
    assert len(plan) == 2048 and plan[0] == (0, 0) and plan[128] == (0, 12) and plan[2047] == (127, 180)
    for k in (0, 5, 191, 400):
        frames = {}
        for s, (loop, off) in enumerate(plan):
            f = bench.frame_index(k, 192, off)
            assert (loop, f) not in frames, (k, s, frames.get((loop, f)))
            frames[(loop, f)] = s
    assert [bench.frame_index(k, 192, 190) for k in range(4)] == [190, 191, 0, 1]
    plan = bench.loop_plan(3584, 128, 192)            # the default: 28 sequences per loop, 6 frames apart
    assert len(plan) == 3584 and plan[128] == (0, 6) and plan[3583] == (127, 162)
    for k in (0, 191, 777):
        shown = {(loop, bench.frame_index(k, 192, off)) for loop, off in plan}
        assert len(shown) == 3584
    seqs, groups = bench.group_sample(3584, 14)
    assert seqs[:14] == [256 * g for g in range(14)] and len(set(groups[:16])) == 14
    seqs, groups = bench.group_sample(2048, 8)
    assert seqs[:8] == [0, 256, 512, 768, 1024, 1280, 1536, 1792] and groups[:8] == list(range(8))
    assert seqs[8:] == [s + 1 for s in seqs[:8]] and len(set(groups)) == 8
    seqs, groups = bench.group_sample(10, 3)          # groups of 4, 3, 3
    assert seqs == [0, 4, 7, 1, 5, 8] and groups == [0, 1, 2, 0, 1, 2]


def test_loop_trajectory_is_closed_and_smooth():
    p = synth.loop_trajectory(192, 3, 0.75)
    assert p.shape == (192, 6)
    step = np.abs(np.diff(np.vstack([p, p[:1]]), axis=0))          # includes the wrap-around step
    assert step[:, :3].max() < 0.03 and np.degrees(step[:, 3:].max()) < 2.0
    assert np.allclose(step[-1], np.abs(p[0] - p[-1]))


@pytest.mark.parametrize("rnd", ["r02", "r03"])
def test_committed_counter_summary_is_reproducible_from_the_csvs(tmp_path, rnd):
    """profiles/r0N_pmc.json (r03: read by bench.py for roofline.binding_roof and valu_issue) is what
    tools/make_pmc_json.py derives from the committed counter and kernel-stats CSVs."""
    import json, shutil, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "profiles")
    for name in ("pmc_sq.csv", "kernel_stats.csv", "pmc.json"):
        shutil.copy(os.path.join(prof, rnd + "_" + name), tmp_path / name)
    out = tmp_path / "out"
    out.mkdir()
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "make_pmc_json.py"), str(tmp_path), str(out),
                           "reclassify"], stdout=subprocess.DEVNULL)
    new = json.load(open(out / "pmc.json"))
    old = json.load(open(os.path.join(prof, rnd + "_pmc.json")))
    assert (new["seqs"], new["groups"]) == (old["seqs"], old["groups"]) == ((2048, 8) if rnd == "r02" else (3584, 14))
    if rnd == "r03":      # the chip-level figure bench.py prints: VALU instructions per tracked frame
        assert abs(new["valu_instructions_per_frame"] - old["valu_instructions_per_frame"]) < 1.0
        assert 1.5e6 < new["valu_instructions_per_frame"] < 2.5e6
    assert set(new["kernels"]) == set(old["kernels"]) and "sia_gn_kernel" in new["kernels"]
    for k, v in new["kernels"].items():
        assert v["bound"] == old["kernels"][k]["bound"]
        assert abs(v["frac"] - old["kernels"][k]["frac"]) < 1e-9
    assert new["kernels"]["sia_gn_kernel"]["bound"] == "latency"


def test_timeline_tool_on_a_synthetic_trace(tmp_path):
    """tools/timeline.py (the overlap summary behind profiles/r02_timeline_*.txt) on a hand-made
    kernel trace: two queues, every alignment kernel overlapped by the other queue's."""
    import csv
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rows, t = [], 0
    for _ in range(40):
        for q in (1, 2):
            s0 = t + q * 10
            rows.append(dict(Start_Timestamp=s0, End_Timestamp=s0 + 500, Queue_Id=q,
                             Kernel_Name="void svo::sia_gn_kernel<1, 2>(svo::SiaArgs const*, int, int)"))
            rows.append(dict(Start_Timestamp=s0 + 600, End_Timestamp=s0 + 800, Queue_Id=q,
                             Kernel_Name="void svo::klt_track_kernel<32>(svo::KltArgs const*)"))
        t += 1000
    path = tmp_path / "trace.csv"
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "timeline.py"), str(path), "0.5"],
                         capture_output=True, text=True, check=True).stdout
    line = [l for l in out.splitlines() if l.startswith("void sia_gn_kernel<1, 2>")][0].split()
    assert abs(float(line[-4]) - 0.5) < 1e-6           # avg us of the 500 ns kernels
    assert 0.9 < float(line[-1]) <= 1.0                # ~one other kernel in flight the whole time
    assert "queue 1:" in out and "queue 2:" in out and "kernels in flight" in out
