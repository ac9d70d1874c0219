"""N > 1: the batch is sharded by batch index, one process per GPU, no data-path collective.  These CPU tests
run bench.py's sharding / timing-reduction logic with world_size 2 over gloo (no GPU needed): every rank owns
`batch` whole transforms and transforms them with the ENGINE (planner + kernels in the CPU emulation, tests/emu;
both the multi-pass schedule and the team kernel), the union covers [0, world*batch) exactly once, and the only
communication is the barrier + MAX all-reduce of the elapsed time."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


WORKER = textwrap.dedent('''
    import os, sys, json
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, os.path.join(%r, "tests"))
    import oracle_lib as O
    import emu_lib as E
    from bench import shard_range, reduce_max_seconds
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n, per_gpu = 4096, 6
    b0, b1 = shard_range(rank, world, per_gpu)
    # each rank transforms ITS OWN shard with the ENGINE -- the unmodified planner + kernel source running in the CPU
    # emulation (tests/emu), two-pass plan and the team kernel -- never with the oracle; the oracle only checks
    x = O.gen_two_tone(n, b0, b1 - b0).astype(np.complex64)
    X, info = E.emu_fft(x, -1, lds_budget=16384)                       # multi-pass schedule (64 x 64)
    Xt, info_t = E.emu_fft_team(x, -1, log2seats=2, n_xcc=2, threads=16, lds_budget=16384)   # team kernel, 2 "XCDs" of 4 seats
    ok = info[0] == 2 and info_t[0] >= 400
    ref = O.oracle_fft(x.astype(np.complex128), -1, "dit")
    ok &= float(np.linalg.norm(X - ref) / np.linalg.norm(ref)) < 2e-6
    ok &= float(np.linalg.norm(Xt - ref) / np.linalg.norm(ref)) < 2e-6
    for i, b in enumerate(range(b0, b1)):
        f, g = O.two_tone_bins(n, b)
        ok &= abs(X[i, f] - n) < 1e-3 * n and abs(X[i, g] - n / 2) < 1e-3 * n
    owned = torch.zeros(world * per_gpu, dtype=torch.int32)
    owned[b0:b1] = 1
    dist.all_reduce(owned)  # test-only bookkeeping: who owns which transform
    t = reduce_max_seconds(dist, 0.01 * (rank + 1), torch.device("cpu"))
    if rank == 0:
        print(json.dumps({"ok": bool(ok), "cover": owned.tolist(), "t": t, "world": world}))
    dist.barrier()
    dist.destroy_process_group()
''')


def test_world_size_2_gloo_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["ok"] and r["world"] == 2
    assert r["cover"] == [1] * 12  # every transform owned exactly once
    assert abs(r["t"] - 0.02) < 1e-9  # MAX over ranks


def test_shard_range_is_a_partition():
    sys.path.insert(0, ROOT)
    from bench import shard_range
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            a, b = shard_range(r, world, 512)
            seen.extend(range(a, b))
        assert seen == list(range(512 * world))


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher must start two ranks itself (a torch.distributed.run child, before the
    parent imports torch) and report n_gpus == 2 -- never a silent one-GPU measurement.  --dry-run: the
    ranks meet over gloo and skip the GPU work, which this container does not have."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["gpus_arg"] == 2 and line["last_rank_shard"] == [512, 1024]


def test_bench_refuses_a_world_that_is_not_gpus():
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_bench_pmc_average_reads_rocprofv3_counter_csv(tmp_path):
    """bench.py's live roofline.traffic: the per-dispatch average of one counter for the dominant kernel, from the CSV layout rocprofv3
    --pmc writes (other kernels and other counters in the same file are ignored; no matching row -> None -> the committed figure is used)."""
    import bench
    d = tmp_path / "run" / "host"
    d.mkdir(parents=True)
    (d / "123_counter_collection.csv").write_text(
        '"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name","Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value","Start_Timestamp","End_Timestamp"\n'
        '1,1,0,1,7,7,131072,5,"void fftk::team_quad_kernel<float, 16, 4, 4, 10, 10, 5, 2>(fftk::TeamParams<float>)",512,0,0,240,0,96,"FETCH_SIZE",2500000.0,1,2\n'
        '2,2,0,1,7,7,131072,5,"void fftk::team_quad_kernel<float, 16, 4, 4, 10, 10, 5, 2>(fftk::TeamParams<float>)",512,0,0,240,0,96,"FETCH_SIZE",2600000.0,3,4\n'
        '3,3,0,1,7,7,1024,6,"void fftk::tile_fft_kernel<float, 8, 1, 0, 0, 0, true, 2563, 0>(fftk::TileParams<float>)",512,0,0,120,0,96,"FETCH_SIZE",11.0,5,6\n'
        '4,4,0,1,7,7,131072,5,"void fftk::team_quad_kernel<float, 16, 4, 4, 10, 10, 5, 2>(fftk::TeamParams<float>)",512,0,0,240,0,96,"WRITE_SIZE",9.0,7,8\n')
    assert bench.pmc_average(str(tmp_path), "team_quad_kernel", "FETCH_SIZE") == 2550000.0
    assert bench.pmc_average(str(tmp_path), "team_quad_kernel", "WRITE_SIZE") == 9.0
    assert bench.pmc_average(str(tmp_path), "team_defer_kernel", "FETCH_SIZE") is None
