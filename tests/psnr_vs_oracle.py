"""bf16 TrainStep against the CPU oracle + torch.optim.Adam (the reference's arithmetic) on the same small scene,
batches and draws: held-out PSNR of both final parameter sets (same fp32 evaluator), per seed, the gap of the means
and its standard error.  The CPU runs go to spawned worker processes (they never touch the GPU).  Lives under tests/
because it executes the oracle as the checker (only tests/, smoke() and bench.py's cpu_baseline leg may).

    python tests/psnr_vs_oracle.py [steps=300] [rays=64] [nc=16] [nf=16] [seeds=8] [lr=1e-3] [modes=bf16,fp32] [lr_end=0] [freq=1] [first_seed=0]
Prints one JSON line."""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]      # ROOT = the repository (this file sits in tests/)


def main():
    import torch.multiprocessing as mp
    import oracle_train as OT
    a = sys.argv[1:]
    steps = int(a[0]) if len(a) > 0 else 300
    b = int(a[1]) if len(a) > 1 else 64
    nc = int(a[2]) if len(a) > 2 else 16
    nf = int(a[3]) if len(a) > 3 else 16
    n_seeds = int(a[4]) if len(a) > 4 else 8
    lr = float(a[5]) if len(a) > 5 else 1e-3
    modes = (a[6] if len(a) > 6 else "bf16,fp32").split(",")
    lr_end = float(a[7]) if len(a) > 7 and float(a[7]) > 0 else None
    freq = float(a[8]) if len(a) > 8 else 1.0
    seed0 = int(a[9]) if len(a) > 9 else 0      # seeds first_seed .. first_seed + seeds - 1 (a second, independent sample)
    noise = 0.5
    cores = OT.usable_cores()
    # the oracle's small-batch steps are op-overhead bound (0.09 s per step at 8 threads, 0.12 at 2): few threads per run,
    # one process per seed — and never more threads than the affinity mask admits (an over-subscribed OpenMP pool on a
    # quota-limited box is orders of magnitude slower)
    procs = max(1, min(n_seeds, cores // 2))
    threads = 2
    t0 = time.perf_counter()
    ctx = mp.get_context("spawn")
    pool = ctx.Pool(procs)
    fut = pool.map_async(OT.cpu_run, [(s, steps, b, nc, nf, lr, noise, threads, lr_end, freq) for s in range(seed0, seed0 + n_seeds)])
    gpu = {}
    for mode in modes:
        gpu[mode] = [OT.gpu_run(s, steps, b, nc, nf, lr, noise, mode, lr_end=lr_end, freq=freq) for s in range(seed0, seed0 + n_seeds)]
    t_gpu = time.perf_counter() - t0
    cpu = sorted(fut.get(timeout=3600))
    pool.close()
    wall = time.perf_counter() - t0
    res = {"config": f"{steps} steps x {b} rays x ({nc}+{nf}) samples, Adam lr {lr} -> {lr_end}, scene freq {freq}, noise_std {noise}, {n_seeds} seeds from {seed0}",
           "cpu_oracle_psnr_db": [round(c[1], 3) for c in cpu], "cpu_first_last_loss": [(round(c[2][0], 4), round(c[2][-1], 4)) for c in cpu],
           "wall_s": round(wall, 1), "gpu_part_s": round(t_gpu, 1), "cpu_procs": procs, "threads_per_proc": threads}
    ref = [c[1] for c in cpu]
    for mode in modes:
        ps = [g[0] for g in gpu[mode]]
        diffs = [p - r for p, r in zip(ps, ref)]
        mean = sum(diffs) / len(diffs)
        sd = math.sqrt(sum((x - mean) ** 2 for x in diffs) / max(1, len(diffs) - 1))
        res[mode] = {"psnr_db": [round(p, 3) for p in ps], "gap_db_per_seed": [round(x, 3) for x in diffs],
                     "mean_gap_db": round(mean, 4), "se_db": round(sd / math.sqrt(len(diffs)), 4),
                     "first_last_loss": [(round(g[1][0], 4), round(g[1][-1], 4)) for g in gpu[mode]]}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
