#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X FFT engine (BASELINE.json metric).

A "step" is ONE pass of the hot path over one batch of synthetic input: a
batched 1D complex-to-complex forward FFT of `batch` transforms of length `n`,
device-resident (inputs are already in HBM when the timed region starts),
executed through the C ABI (fft_gpu_execute_ptr on a plan from
fft_gpu_plan_1d_ex).  Default workload = BASELINE.json configs[2], the one the
north-star target is quoted on: N = 1048576 fp32 complex, batch = 512 per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, the batch is sharded by batch index (each rank owns
`batch` whole transforms -> weak scaling), there is NO data-path collective;
torch.distributed (RCCL) is used only for the barriers around the timed region
and the max-over-ranks of the elapsed time.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "fft-implementation-in-c_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"

WORKLOADS = {
    # name: (n, batch per GPU, dtype, description)
    "1m": (1 << 20, 512, "f32", "N=1048576 fp32 complex, batch=512 per GPU (BASELINE configs[2], north-star)"),
    "64k": (1 << 16, 4096, "f32", "N=65536 fp32 complex, batch=4096 per GPU (BASELINE configs[1])"),
    "256k": (1 << 18, 1024, "f32", "N=262144 fp32 complex, batch=1024 per GPU (BASELINE configs[3] shard at 8 GPUs)"),
    "prime": (1000003, 64, "f64", "N=1000003 (prime) fp64 complex via Bluestein, batch=64 (BASELINE configs[4])"),
    "1k": (1024, 65536, "f32", "N=1024 fp32 complex, batch=65536 (single-pass LDS kernel)"),
}


def shard_range(rank, world, batch_per_gpu):
    """Weak scaling by batch index: rank r owns transforms [r*batch, (r+1)*batch) of the world*batch job."""
    return rank * batch_per_gpu, (rank + 1) * batch_per_gpu


def reduce_max_seconds(dist, seconds, device):
    """The only collective of the benchmark: MAX over ranks of the elapsed time (never on the data path)."""
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def make_input(torch, n, batch, dtype, b0, device):
    """Closed-form two-tone complex sinusoids: X_b[f_b] = N, X_b[g_b] = N/2, 0 elsewhere."""
    cdtype = torch.complex64 if dtype == "f32" else torch.complex128
    out = torch.empty((batch, n), dtype=cdtype, device=device)
    j = torch.arange(n, dtype=torch.int64, device=device)
    slab = max(1, min(batch, (1 << 24) // n))
    for s in range(0, batch, slab):
        bb = torch.arange(b0 + s, b0 + min(batch, s + slab), dtype=torch.int64, device=device)
        f = (1 + 7 * bb) % n
        g = (n // 3 + 13 * bb) % n
        g = torch.where(g == f, (g + 1) % n, g)
        pf = (f[:, None] * j[None, :]) % n
        pg = (g[:, None] * j[None, :]) % n
        af = pf.to(torch.float64) * (2.0 * 3.14159265358979323846 / n)
        ag = pg.to(torch.float64) * (2.0 * 3.14159265358979323846 / n)
        re = torch.cos(af) + 0.5 * torch.cos(ag)
        im = torch.sin(af) + 0.5 * torch.sin(ag)
        out[s:s + bb.numel()] = torch.complex(re, im).to(cdtype)
    return out


def cpu_baseline(n, dtype, budget_s=12.0):
    """Time the reference's own CPU path on this host, on a bounded sample of the same workload.

    kind "reference": radix2_dit_fft (algorithms/core/radix2_dit.c:59-120, what fft_auto runs for every
    power of two, as split_radix/radix4 are the same loop) from oracle/_ref/libref_fast.so, i.e. the real
    reference sources built with its shipped optimisation flags (Makefile:7) at a portable ISA level.
    The reference is fp64-only (complex_t = double complex); all host cores, one transform per thread at a
    time (task parallelism over the batch index, optimizations/parallel_fft.c:424-427).
    Falls back to kind "port" (oracle/liboracle_fast.so, the restatement) only if _ref is absent.
    """
    import numpy as np
    import oracle_lib as O
    kind, fn, blu = None, None, None
    ref_fast = os.path.join(ROOT, "oracle", "_ref", "libref_fast.so")
    pow2 = (n & (n - 1)) == 0
    if os.path.exists(ref_fast):
        lib = C.CDLL(ref_fast)
        f = lib.radix2_dit_fft if pow2 else lib.bluestein_fft
        f.argtypes = [C.c_void_p, C.c_int, C.c_int]
        f.restype = None
        kind = "reference"

        def fn(ptr):
            f(ptr, n, -1)
    else:
        import subprocess
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle_fast.so"], check=True)
        lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle_fast.so"))
        lib.oracle_fft_batch.argtypes = [C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_int]
        kind = "port"
        algo = 0 if pow2 else 4

        def fn(ptr):
            lib.oracle_fft_batch(ptr, n, 1, -1, algo)
    cores = os.cpu_count() or 1
    threads = max(1, min(cores, 64))
    x0 = O.gen_two_tone(n, 0, 1)[0]
    # calibrate on one thread
    a = x0.copy()
    t0 = time.perf_counter()
    fn(a.ctypes.data)
    t_one = time.perf_counter() - t0
    reps = max(1, int(budget_s * 0.6 / max(t_one, 1e-6)))
    reps = min(reps, 200)
    single = []
    for _ in range(min(reps, 5)):
        a = x0.copy()
        t0 = time.perf_counter()
        fn(a.ctypes.data)
        single.append(time.perf_counter() - t0)
    t_single = min(single)
    per_thread = max(1, min(reps, int(budget_s * 0.5 / max(t_one, 1e-6))))
    bufs = [x0.copy() for _ in range(threads)]
    barrier = threading.Barrier(threads + 1)

    def work(buf):
        barrier.wait()
        for _ in range(per_thread):
            buf[:] = x0
            fn(buf.ctypes.data)
        barrier.wait()

    th = [threading.Thread(target=work, args=(b,)) for b in bufs]
    for t in th:
        t.start()
    barrier.wait()
    t0 = time.perf_counter()
    barrier.wait()
    wall = time.perf_counter() - t0
    for t in th:
        t.join()
    total_pts = float(n) * threads * per_thread
    out = {
        "value": total_pts / wall / 1e9, "unit": "Gpoint/s", "cores": threads, "kind": kind,
        "sample": "%d transforms of N=%d fp64 (%s, the reference's only precision) per thread on %d threads, "
                  "two-tone input, forward; 1-thread best: %.4f Gpoint/s (%.1f ms/transform)"
                  % (per_thread, n, "radix2_dit_fft" if pow2 else "bluestein_fft", threads, n / t_single / 1e9, t_single * 1e3),
        "single_thread_value": n / t_single / 1e9,
    }
    if pow2:
        out["fp32_comparators"] = cpu_fp32_comparators(n, threads)
    return out


def cpu_fp32_comparators(n, threads):
    """The two fp32 CPU comparators the north star names, timing only (a few seconds):
    - optimizations/simd_fft.c `fft_radix2_sse2` from oracle/_ref/libref_simd.so -- the reference's SSE2 SoA path;
      its results are numerically wrong (one twiddle for four butterflies), so it is a TIMING
      comparator only;
    - the oracle's fp32 interleaved radix-2 DIT with correct twiddles, all cores via OpenMP over the batch index
      (oracle/liboracle_fast.so, built here with the reference's shipped flags)."""
    import numpy as np
    res = {}
    try:
        simd = os.path.join(ROOT, "oracle", "_ref", "libref_simd.so")
        if os.path.exists(simd):
            lib = C.CDLL(simd)
            lib.allocate_simd_complex.restype = C.c_void_p
            lib.allocate_simd_complex.argtypes = [C.c_int]
            lib.fft_radix2_sse2.argtypes = [C.c_void_p, C.c_int, C.c_int]
            lib.free_simd_complex.argtypes = [C.c_void_p]
            h = lib.allocate_simd_complex(n)
            ptrs = (C.c_void_p * 2).from_address(h)
            re = np.ctypeslib.as_array((C.c_float * n).from_address(ptrs[0]))
            im = np.ctypeslib.as_array((C.c_float * n).from_address(ptrs[1]))
            best = 1e9
            import oracle_lib as O
            x0 = O.gen_two_tone(n, 0, 1, np.complex64)[0]
            for _ in range(5):
                re[:] = x0.real
                im[:] = x0.imag
                t0 = time.perf_counter()
                lib.fft_radix2_sse2(h, n, -1)
                best = min(best, time.perf_counter() - t0)
            lib.free_simd_complex(h)
            res["reference_simd_fft_sse2_1thread"] = {"value": n / best / 1e9, "unit": "Gpoint/s", "cores": 1,
                                                       "note": "optimizations/simd_fft.c:143-230, timing only (results wrong in the reference)"}
    except Exception as e:
        res["reference_simd_fft_sse2_1thread"] = {"value": None, "note": "failed: %r" % (e,)}
    try:
        import subprocess
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle_fast.so"], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle_fast.so"))
        lib.oracle_fft_batch_f32.argtypes = [C.c_void_p, C.c_int, C.c_long, C.c_int]
        batch = max(threads, 8)
        import oracle_lib as O
        x0 = O.gen_two_tone(n, 0, batch, np.complex64)
        x = x0.copy()
        lib.oracle_fft_batch_f32(x.ctypes.data, n, batch, -1)
        x[:] = x0
        t0 = time.perf_counter()
        lib.oracle_fft_batch_f32(x.ctypes.data, n, batch, -1)
        dt = time.perf_counter() - t0
        res["port_fp32_radix2_dit_all_cores"] = {"value": float(n) * batch / dt / 1e9, "unit": "Gpoint/s", "cores": threads,
                                                  "note": "oracle_radix2_dit_f32, OpenMP over %d transforms" % batch}
    except Exception as e:
        res["port_fp32_radix2_dit_all_cores"] = {"value": None, "note": "failed: %r" % (e,)}
    try:
        # the SoA 4-wide variant SURVEY.md 8d asks for: optimizations/simd_fft.c's structure (split real[] / imag[], four
        # butterflies per step, :143-230) with a correct twiddle per lane -- oracle_radix2_soa4_f32
        lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle_fast.so"))
        lib.oracle_radix2_soa4_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        lib.oracle_soa4_batch_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_int]
        import oracle_lib as O
        batch = max(threads, 8)
        x0 = O.gen_two_tone(n, 0, batch, np.complex64)
        re0, im0 = np.ascontiguousarray(x0.real), np.ascontiguousarray(x0.imag)
        best1 = 1e9
        for _ in range(3):
            re, im = re0[0].copy(), im0[0].copy()
            t0 = time.perf_counter()
            lib.oracle_radix2_soa4_f32(re.ctypes.data, im.ctypes.data, n, -1)
            best1 = min(best1, time.perf_counter() - t0)
        re, im = re0.copy(), im0.copy()
        lib.oracle_soa4_batch_f32(re.ctypes.data, im.ctypes.data, n, batch, -1)
        re[:], im[:] = re0, im0
        t0 = time.perf_counter()
        lib.oracle_soa4_batch_f32(re.ctypes.data, im.ctypes.data, n, batch, -1)
        dt = time.perf_counter() - t0
        res["port_fp32_soa4_radix2_1thread"] = {"value": n / best1 / 1e9, "unit": "Gpoint/s", "cores": 1,
                                                "note": "oracle_radix2_soa4_f32: simd_fft.c's SoA 4-wide structure, correct per-lane twiddles"}
        res["port_fp32_soa4_radix2_all_cores"] = {"value": float(n) * batch / dt / 1e9, "unit": "Gpoint/s", "cores": threads,
                                                  "note": "the same, OpenMP over %d transforms" % batch}
    except Exception as e:
        res["port_fp32_soa4_radix2_1thread"] = {"value": None, "note": "failed: %r" % (e,)}
    return res


def kernel_source_hash():
    """Identity of the kernels a PMC profile was taken on: sha256 over the kernel sources.  profiles/pmc_traffic.json
    entries carry it; a traffic figure is attached to the bench line only when it still matches (the
    committed figure must not silently outlive the kernel it was measured on)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "fft-implementation-in-c_amd", "csrc")
    for f in ("fft_device.h", "fft_codelets.h", "fft_kernels.h", "fft_team.h", "fft_team_defer.h", "fft_team_list.h", "fft_team_quad.h",
              "fft_team_quad_decl.h", "fft_team_quad.hip"):
        with open(os.path.join(d, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_average(out_dir, kernel_substr, counter):
    """Average of `counter` over the dispatches of the kernels whose name contains `kernel_substr`, from the counter_collection CSVs
    rocprofv3 --pmc leaves under out_dir (columns Kernel_Name, Counter_Name, Counter_Value); None when there is no such row."""
    import csv
    import glob
    tot, cnt = 0.0, 0
    for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if kernel_substr in row["Kernel_Name"] and row["Counter_Name"] == counter:
                    tot += float(row["Counter_Value"])
                    cnt += 1
    return tot / cnt if cnt else None


def live_pmc_traffic(workload, kernel_substr):
    """roofline.traffic measured in THIS run (round-2 review: the committed figure is builder-side): two child runs of this command under
    `rocprofv3 --pmc` -- counters only, one counter per pass, never combined with tracing -- and the average of the dominant kernel's
    dispatches.  Bytes across the L2's memory side per launch: FETCH_SIZE (KB, 128-byte requests tallied at 64: x 2, MI355X_MICROARCH.md HBM
    section) + WRITE_SIZE (KB).  Returns (bytes, read bytes, written bytes) or None when rocprofv3 is missing or a pass fails."""
    import shutil
    import subprocess
    import tempfile
    def skipped(why):  # (the committed figure is attached instead; say so where a reader of the log will see it)
        print("bench.py: live PMC traffic not measured: %s" % why, file=sys.stderr, flush=True)
        return None
    tool = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(tool):
        return skipped("no rocprofv3")
    # never from inside a profiled run: a profiler's preloaded library is inherited by the child, and starting a second profiler from
    # there is refused by the GPU boxes' exec guard (and would profile the profiler)
    if "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.upper().startswith(("ROCPROF", "ROCP_", "ROCTX")) for k in os.environ):
        return skipped("this process runs under a profiler")
    vals = {}
    for cset in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="fft_pmc_", dir="/tmp")
        try:
            cmd = [tool, "--pmc", cset, "-d", d, "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__),
                   "--workload", workload, "--steps", "3", "--warmup", "1", "--no-check", "--no-cpu-baseline", "--no-secondary", "--no-live-traffic"]
            env = dict(os.environ)
            env["TMPDIR"] = "/tmp"
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
                env.pop(k, None)
            r = subprocess.run(cmd, cwd="/tmp", env=env, timeout=420, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            if r.returncode != 0:
                return skipped("the %s pass exited with %d: %s" % (cset, r.returncode, r.stderr.decode(errors="replace")[-300:]))
            avg = pmc_average(d, kernel_substr, cset)
            if avg is None:
                return skipped("no %s row for %s" % (cset, kernel_substr))
            vals[cset] = avg
        except Exception as e:
            return skipped("%r" % (e,))
        finally:
            shutil.rmtree(d, ignore_errors=True)
    rd, wr = vals["FETCH_SIZE"] * 2048.0, vals["WRITE_SIZE"] * 1024.0
    return rd + wr, rd, wr


def device_copy_gbs(torch, x, y, iters=10):
    """What a plain device copy of the same buffers achieves on THIS box (read + written bytes per second): the
    practical ceiling the guide quotes at 6.29 TB/s; roofline.frac_of_copy = achieved / this."""
    y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            y.copy_(x)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return 2.0 * x.numel() * x.element_size() / (best * 1e-3) / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="1m", choices=sorted(WORKLOADS))
    ap.add_argument("--algo", default="auto")
    ap.add_argument("--inplace", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--settle-ms", type=float, default=200.0,
                    help="untimed executes BEFORE the W warmup steps until this much wall time has passed: a fresh process starts at idle "
                         "clocks and the first few ms-sized steps would otherwise be timed on the ramp (0 = off)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the fp64 2^19 secondary line and the device copy")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic live (two rocprofv3 --pmc child runs of this command); use profiles/pmc_traffic.json")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="(internal) print the CPU baseline JSON for --workload and exit")
    ap.add_argument("--dry-run", action="store_true",
                    help="(tests) launcher rehearsal without a GPU: the ranks meet over gloo, reduce a dummy time and rank 0 prints n_gpus")
    args = ap.parse_args()
    if args.cpu_baseline_only:  # child process of the main run: a crash here (e.g. SIGILL in a comparator) cannot lose the GPU line
        n, batch, dtype, desc = WORKLOADS[args.workload]
        print(json.dumps(cpu_baseline(n, dtype)), flush=True)
        return

    # --gpus N without a launcher around us: start the N ranks OURSELVES, as a torch.distributed.run child process, BEFORE
    # anything in this process imports torch or touches a GPU (a process that has initialised the GPU must never exec or
    # re-exec; this parent only waits and hands the child's exit code on).  `python bench.py --gpus 8` therefore can never
    # silently measure one GPU: the ranks below insist on WORLD_SIZE == --gpus and on N visible devices.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or let bench.py start its own ranks: "
                         "run it without a launcher)" % (args.gpus, world, args.gpus))
    if args.dry_run:
        import torch
        import torch.distributed as dist_mod
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist_mod.init_process_group(backend="gloo")
            dist_mod.barrier()
            t = torch.tensor([1.0 + rank], dtype=torch.float64)
            dist_mod.all_reduce(t, op=dist_mod.ReduceOp.MAX)
            assert float(t.item()) == float(world)
            dist_mod.barrier()
            dist_mod.destroy_process_group()
        if rank == 0:
            b0, b1 = shard_range(world - 1, world, WORKLOADS[args.workload][1])
            print(json.dumps({"dry_run": True, "n_gpus": world, "gpus_arg": args.gpus, "last_rank_shard": [b0, b1]}), flush=True)
        return

    import numpy as np
    import torch
    import fftlib

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    if torch.cuda.device_count() < world or local_rank >= torch.cuda.device_count():
        raise SystemExit("--gpus %d but only %d device(s) are visible" % (args.gpus, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)

    n, batch, dtype, desc = WORKLOADS[args.workload]
    npdtype = np.complex64 if dtype == "f32" else np.complex128
    esz = 8 if dtype == "f32" else 16

    lib = fftlib.init()
    plan = fftlib.Plan(n, batch, fftlib.FFT_FORWARD, npdtype, fftlib.ALGO_NAMES[args.algo])
    stream = torch.cuda.current_stream(device)
    plan.set_stream(stream.cuda_stream or fftlib.HIP_STREAM_LEGACY)  # handle 0 (torch's default stream) would mean "the plan's own stream"
    info = plan.info()

    b_first, b_last = shard_range(rank, world, batch)
    x = make_input(torch, n, batch, dtype, b_first, device)
    y = x if args.inplace else torch.empty_like(x)
    if args.inplace:
        x_keep = None
    torch.cuda.synchronize()

    def step():
        plan.execute_ptr(x.data_ptr(), y.data_ptr())

    # ---- correctness of what is being timed: analytic spectrum of every transform (first execute)
    check = {}
    if not args.no_check:
        src = x.clone() if args.inplace else x
        step()
        torch.cuda.synchronize()
        Y = y
        bb = torch.arange(rank * batch, rank * batch + batch, dtype=torch.int64, device=device)
        f = (1 + 7 * bb) % n
        g = (n // 3 + 13 * bb) % n
        g = torch.where(g == f, (g + 1) % n, g)
        rows = torch.arange(batch, device=device)
        pf = Y[rows, f]
        pg = Y[rows, g]
        peak_err = max(float((pf - n).abs().max()) / n, float((pg - n / 2).abs().max()) / n)
        tot = torch.linalg.vector_norm(Y.to(torch.complex128) if dtype == "f32" and batch * n <= (1 << 26) else Y, dim=1)
        want = float(n) * (1.25 ** 0.5)
        norm_err = float(((tot - want).abs() / want).max())
        check = {"peak_rel_err": peak_err, "norm_rel_err": norm_err, "transforms_checked": batch}
        tol = 1e-4 if dtype == "f32" else 1e-6
        if not (peak_err < tol and norm_err < tol):
            raise SystemExit("bench: result check FAILED %r" % (check,))
        if args.inplace:
            x.copy_(src)
            del src

    if args.settle_ms > 0:  # clocks / power state settle (untimed, in front of the warmup the contract asks for)
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
            step()
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
        if args.inplace:
            pass
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        elapsed = reduce_max_seconds(dist, elapsed, device)

    # ---- device-side duration of the same K steps, HIP events on the stream the kernels run on
    ev_ms = plan.timed(x.data_ptr(), y.data_ptr(), args.steps)
    torch.cuda.synchronize()
    # per-step device times (one HIP-event pair per step): median and min next to the mean of the timed region
    step_ms = sorted(plan.timed(x.data_ptr(), y.data_ptr(), 1) for _ in range(max(5, min(args.steps, 50))))
    torch.cuda.synchronize()
    # which schedule ran?  0: the one-round-trip team kernel (fft_team.h) did the work; 1: its XCD teams could not be
    # formed and the multi-pass fallback queued behind it did; -1: the plan has no team kernel (multi-pass schedule)
    team_status = plan.team_status()
    team = info.team_tiles > 0 and team_status == 0
    team_kernel_name = {1: "team_fft_kernel", 2: "team_defer_kernel", 3: "team_quad_kernel"}.get(info.team_kernel) if team else None
    # ---- per-launch kernel durations, one more execute with a HIP event after every launch
    per_pass = []
    if not info.bluestein_m:
        prof = [plan.profile_passes(x.data_ptr(), y.data_ptr(), 6) for _ in range(3)][-1]
        team_name = {1: "team_fft_kernel", 2: "team_defer_kernel", 3: "team_quad_kernel"}.get(info.team_kernel, "team kernel")
        names = ([team_name] + ["fallback pass %d (returns at once)" % i for i in range(8)]) if team else \
                ["tile_fft_kernel pass %d" % i for i in range(8)]
        per_pass = [{"pass": i, "kernel": names[i], "launches_per_step": c, "avg_launch_ms": m / c, "ms_per_step": m}
                    for i, (m, c) in enumerate(prof)]
    torch.cuda.synchronize()

    points_per_step = float(n) * batch * world
    bytes_alg_per_step_gpu = 2.0 * n * batch * esz  # read every input once + write every output once
    ms_per_step = elapsed / args.steps * 1e3
    value = points_per_step / (elapsed / args.steps) / 1e9
    ev_ms_per_step = ev_ms / args.steps
    achieved = bytes_alg_per_step_gpu / (ev_ms_per_step * 1e-3) / 1e9
    n_groups = -(-batch // max(1, info.chunk_batch)) if info.n_passes > 1 else 1
    if info.bluestein_m:  # two transforms of length m per group; fused: 1 no element-wise kernels, 2 and the middle two passes as one
        launches = ((2 * max(1, info.n_passes) - (1 if info.fused == 2 else 0)) * n_groups) + (0 if info.fused else 3)
    else:
        launches = info.n_passes * n_groups
    units_per_launch = min(batch, info.chunk_batch) if info.n_passes > 1 else batch
    if team:
        # ONE team_fft_kernel launch carries the whole batch: every transform is read from HBM once and written once
        # (the four-step transposition stays inside the XCDs), so the launch's own duration is the roofline denominator
        n_groups, units_per_launch = 1, batch
        team_ms = per_pass[0]["avg_launch_ms"] if per_pass else ev_ms_per_step
        achieved = bytes_alg_per_step_gpu / (team_ms * 1e-3) / 1e9
        tile_points = 8192 if dtype == "f32" else 4096  # one 64 KiB tile
        team_cus = max(1, n // (info.team_tiles * tile_points))
        team_src = {1: "csrc/fft_team.h team_fft_kernel", 2: "csrc/fft_team_defer.h team_defer_kernel",
                    3: "csrc/fft_team_quad.h team_quad_kernel (256-byte row segments, both steps decimated in time by 4, the "
                       "exchange between them in four rounds through the XCD's L2; n = 2^18 ... 2^20: one image per seat in the window, per-seat counters)"}.get(info.team_kernel, "team kernel")
        kernel_desc = ("%s: ONE launch per step transforms all %d transforms, a whole transform per team of "
                       "%d CUs of one XCD (256 workgroups = %d teams, %d 64 KiB tiles per workgroup and step); the %d multi-pass launches "
                       "queued behind it as its fallback return at once" % (team_src, batch, team_cus, 256 // team_cus, info.team_tiles, launches))
    else:
        kernel_desc = ("tile_fft_kernel: one launch per pass per group of %d transforms (%d launches per step); the "
                       "dominant unit of work is the launch SET that carries a group through all %d passes"
                       % (units_per_launch, launches, max(1, info.n_passes)))
        if info.bluestein_m:
            kernel_desc += ("; Bluestein = forward + inverse transform of length %d with the chirp / spectral products fused into "
                            "their first load and last store%s" % (info.bluestein_m, ", the forward's last pass and the inverse's first "
                            "as ONE kernel (tile_fft_ba_kernel)" if info.fused == 2 else ""))

    achieved_wall = bytes_alg_per_step_gpu / (ms_per_step * 1e-3) / 1e9
    # HBM traffic from the PMC counters comes from separate rocprofv3 --pmc runs (never combined with tracing);
    # the committed summary is attached when it was taken on this same workload / plan shape.
    traffic, traffic_note = None, "collected by separate rocprofv3 --pmc runs, see profiles/"
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pt = json.load(f)
        ent = pt.get(args.workload + ("_team" if team else ""))
        if ent and ent.get("factors") == [v for v in info.factors if v] and ent.get("units_per_launch") == units_per_launch:
            if ent.get("kernel_source_hash") == kernel_source_hash():
                traffic = ent["hbm_bytes_per_launch_set"]
                traffic_note = ent["note"]
            else:
                traffic_note = ("profiles/pmc_traffic.json holds a figure for this workload, taken on other kernel sources (hash %s, now %s): "
                                "not attached; re-run tools/profile_round.sh" % (ent.get("kernel_source_hash"), kernel_source_hash()))
    except Exception:
        pass

    traffic_source = "profiles/pmc_traffic.json (committed rocprofv3 --pmc summary, same kernel sources)" if traffic is not None else None
    traffic_committed = traffic
    if rank == 0 and world == 1 and team and team_kernel_name and not args.no_live_traffic:
        live = live_pmc_traffic(args.workload, team_kernel_name)
        if live:
            traffic = live[0]
            traffic_source = "live: two rocprofv3 --pmc child runs of this command (FETCH_SIZE x 2 + WRITE_SIZE, average per launch of %s)" % team_kernel_name
            traffic_note = ("measured in this run: reads %.2f GB, writes %.2f GB per launch against %.2f GB algorithmic = %.2f x; committed summary: %s"
                            % (live[1] / 1e9, live[2] / 1e9, 2.0 * n * esz * units_per_launch / 1e9, live[0] / (2.0 * n * esz * units_per_launch),
                               ("%.2f GB" % (traffic_committed / 1e9)) if traffic_committed else "none for these sources"))

    result = {
        "metric": "Gpoint/s + achieved HBM GB/s, batched 1D c2c FFT at 1/2/4/8 MI355X",
        "value": value, "unit": "Gpoint/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "ms_per_step_median": step_ms[len(step_ms) // 2], "ms_per_step_min": step_ms[0],
        "value_median": float(n) * batch * world / (step_ms[len(step_ms) // 2] * 1e-3) / 1e9,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic",
        "achieved_hbm_gbs": 2.0 * n * batch * world * esz / (elapsed / args.steps) / 1e9,
        "config": {
            "workload": desc, "n": n, "batch_per_gpu": batch, "batch_total": batch * world,
            "direction": "forward", "placement": "in-place" if args.inplace else "out-of-place",
            "parallelism": "batch index sharded over %d GPU(s), one process per GPU, no collectives" % world,
            "algo": args.algo, "passes": info.n_passes, "factors": [v for v in info.factors if v],
            "chunk_batch": info.chunk_batch, "bluestein_m": info.bluestein_m,
            "schedule": "team kernel: one HBM round trip, a whole transform per team of CUs of one XCD" if team else
                        ("multi-pass (team kernel fell back)" if team_status == 1 else "multi-pass"),
            "team_tiles": info.team_tiles, "team_status": team_status,
            "device": lib.fft_gpu_get_device_name().decode(),
        },
        "roofline": {
            # achieved / frac: from ms_per_step, the wall clock of the timed region (what the driver can check against its own clock);
            # achieved_kernel / frac_kernel: from the dominant kernel's own HIP-event time (the figure rocprofv3's average must agree with)
            "bound": "hbm", "achieved": achieved_wall, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved_wall / HBM_PEAK_GBS, "achieved_kernel": achieved, "frac_kernel": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_source,
            "traffic_over_algorithmic": (traffic / (2.0 * n * esz * units_per_launch)) if traffic else None,
            "dominant_kernel": team_kernel_name if team else ("tile_fft_ba_kernel + tile_fft_kernel" if info.bluestein_m else
                                                                 ("wide_row_kernel / tile_fft_kernel (single pass)" if info.n_passes <= 1 else "tile_fft_kernel")),
            "kernel": kernel_desc,
            "algorithmic_bytes_per_unit": 2.0 * n * esz,
            "units_per_launch_set": units_per_launch,
            "algorithmic_bytes_per_launch_set": 2.0 * n * esz * units_per_launch,
            "launch_set_ms": (per_pass[0]["avg_launch_ms"] if (team and per_pass) else ev_ms_per_step / n_groups),
            "per_pass": per_pass,
            "hip_event_ms_per_step": ev_ms_per_step,
            "note": "achieved = 2*N*sizeof(complex) bytes per transform x transforms per step / ms_per_step (the timed region's wall clock, rank 0's "
                    "share); achieved_kernel = the same bytes per launch set / HIP-event duration of the set (events on the plan's stream; team "
                    "schedule: of the team kernel's launch alone); per_pass = live HIP-event time of each launch; "
                    "traffic: " + traffic_note,
        },
        "check": check,
    }
    if not args.no_secondary:
        # the practical ceiling on this box, and an fp64 line (N = 2^19 is the largest fp64 team-kernel size)
        try:
            copy_torch = device_copy_gbs(torch, x, y if not args.inplace else torch.empty_like(x))
            # hand-written streams over 1 GiB, best of 12-13 launch shapes each (with / without the non-temporal hint, 1 / 4 / 8
            # accesses in flight per thread, 8 / 16 workgroups per CU; copy: also an LDS-DMA tile copy in the engine's own shape)
            copy_own = lib.fft_gpu_stream_bench_hip(1 << 30, 5, 0)
            copy = max(copy_torch, copy_own)
            result["roofline"]["copy_gbs"] = copy
            result["roofline"]["copy_gbs_torch"] = copy_torch
            result["roofline"]["copy_gbs_kernel"] = copy_own
            result["roofline"]["read_gbs"] = lib.fft_gpu_stream_bench_hip(1 << 30, 5, 1)
            result["roofline"]["write_gbs"] = lib.fft_gpu_stream_bench_hip(1 << 30, 5, 2)
            result["roofline"]["frac_of_copy"] = achieved_wall / copy
        except Exception as e:
            result["roofline"]["copy_gbs"] = None
            result["roofline"]["copy_note"] = "failed: %r" % (e,)
        if world == 1 and args.workload == "1m":
            try:
                del x, y
                torch.cuda.empty_cache()
                n2, b2 = 1 << 19, 256
                p2 = fftlib.Plan(n2, b2, fftlib.FFT_FORWARD, np.complex128)
                p2.set_stream(stream.cuda_stream or fftlib.HIP_STREAM_LEGACY)
                x2 = make_input(torch, n2, b2, "f64", 0, device)
                y2 = torch.empty_like(x2)
                p2.timed(x2.data_ptr(), y2.data_ptr(), 2)
                t2 = sorted(p2.timed(x2.data_ptr(), y2.data_ptr(), 1) for _ in range(10))
                st2 = p2.team_status()
                bb = torch.arange(b2, dtype=torch.int64, device=device)
                pk = y2[torch.arange(b2, device=device), (1 + 7 * bb) % n2]
                ok2 = float((pk - n2).abs().max()) / n2 < 1e-6
                med = t2[len(t2) // 2]
                result["secondary_fp64"] = {
                    "workload": "N=524288 fp64 complex, batch=256", "value": float(n2) * b2 / (med * 1e-3) / 1e9, "unit": "Gpoint/s",
                    "ms_per_step_median": med, "ms_per_step_min": t2[0], "achieved_gbs": 2.0 * n2 * b2 * 16 / (med * 1e-3) / 1e9,
                    "frac": 2.0 * n2 * b2 * 16 / (med * 1e-3) / 1e9 / HBM_PEAK_GBS, "team_status": st2, "check_ok": ok2}
                p2.destroy()
                del x2, y2
            except Exception as e:
                result["secondary_fp64"] = {"value": None, "note": "failed: %r" % (e,)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # in a child process: the comparators load CPU libraries built on another host; whatever happens to them, the
        # GPU line below is printed
        try:
            import subprocess
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload", args.workload],
                                 capture_output=True, text=True, timeout=600)
            result["cpu_baseline"] = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
            result["cpu_baseline"] = {"value": None, "unit": "Gpoint/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    elif rank == 0:
        result["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(result), flush=True)
    plan.destroy()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
