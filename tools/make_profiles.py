"""Condense gpurun_out/prof_<tag>_c<config>/ (tools/collect_profiles.sh) into the files committed under profiles/:
    python tools/make_profiles.py r02 2
writes profiles/<tag>_kernel_stats[_configC].csv, <tag>_pmc_per_kernel[_configC].csv, <tag>_bench_line[_configC].json and
<tag>_traffic_configC.json (HBM bytes per step and per launch from the PMC passes, read back by bench.py)."""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
PMC_STEPS = 5            # tools/collect_profiles.sh: --steps 3 --warmup 2 --repeats 1, eager
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_c{cfg}")
dst = os.path.join(ROOT, "profiles")
sfx = "" if cfg == 2 else f"_config{cfg}"


try:
    import subprocess
    GIT_SHA = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except OSError:
    GIT_SHA = None


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    if not hits:
        raise SystemExit(f"missing {pattern} under {src}")
    return hits[0]


shutil.copy(one("stats/**/*kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats{sfx}.csv"))
shutil.copy(os.path.join(src, "bench_line.json"), os.path.join(dst, f"{tag}_bench_line{sfx}.json"))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
EXTRA = ["SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT", "SQ_ACTIVE_INST_LDS",
         "SQ_INST_CYCLES_VMEM_WR", "SQ_INST_CYCLES_VMEM_RD", "SQ_BUSY_CU_CYCLES", "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU_MFMA_MOPS_BF16",
         "SQ_LDS_DATA_FIFO_FULL", "SQ_LDS_CMD_FIFO_FULL", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_MFMA"]
subs = ["pmc_fetch", "pmc_write", "pmc_sq"] + [s_ for s_ in ("pmc_lds", "pmc_vmem", "pmc_issue")
                                               if glob.glob(os.path.join(src, f"{s_}/**/*counter_collection.csv"), recursive=True)]
for sub in subs:
    for r in csv.DictReader(open(one(f"{sub}/**/*counter_collection.csv"))):
        k = r["Kernel_Name"]
        name = k.split("(")[0].replace("void ", "")
        agg[(name, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES",
        "SQ_WAIT_ANY"] + (EXTRA if len(subs) > 3 else [])
step_bytes, per_kernel_bytes, per_kernel_n = 0.0, collections.defaultdict(float), collections.defaultdict(float)
with open(os.path.join(dst, f"{tag}_pmc_per_kernel{sfx}.csv"), "w") as f:
    f.write("kernel,grid,launches,FETCH_SIZE_KB,fetch_bytes_corrected,WRITE_SIZE_KB,write_bytes," + ",".join(cols[2:]) +
            ",mfma_busy_share\n")
    for (name, grid), c in sorted(agg.items()):
        avg = {k: sum(v) / len(v) for k, v in c.items()}
        n = max(len(v) for v in c.values())
        fk, wk = avg.get("FETCH_SIZE", 0.0), avg.get("WRITE_SIZE", 0.0)
        # FETCH_SIZE / WRITE_SIZE count KiB; gfx950 tallies a wide coalesced read at half its bytes (MI355X_MICROARCH.md)
        fb, wb = fk * 1024 * 2, wk * 1024
        step_bytes += (fb + wb) * n / PMC_STEPS
        base = name.split("<")[0]
        per_kernel_bytes[base] += (fb + wb) * n
        per_kernel_n[base] += n
        # share of the matrix pipes' cycles that issue MFMAs: BUSY_CYCLES summed over 1024 SIMDs / (8 XCD-summed clock)
        share = ""
        if avg.get("SQ_VALU_MFMA_BUSY_CYCLES") and avg.get("GRBM_GUI_ACTIVE"):
            share = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (avg["GRBM_GUI_ACTIVE"] / 8.0)
        if not any(s in name for s in ("hn_", "Cijk", "at::", "elementwise", "reduce")):
            continue
        f.write(f"\"{name}\",{grid},{n},{fk},{fb},{wk},{wb}," + ",".join(str(avg.get(k, "")) for k in cols[2:]) +
                f",{share}\n")
line = json.load(open(os.path.join(src, "bench_line.json")))
traffic = {"config": cfg, "rays": line["config"]["rays_per_gpu"], "nc": line["config"]["n_samples"],
           "nf": line["config"]["n_importance"], "dtype": line.get("dtype", "bf16"), "pmc_steps": PMC_STEPS,
           "bytes_per_step": step_bytes,
           # the kernels these counters were collected on (bench.py refuses to quote the file for any other build)
           "build": dict(line.get("build") or {}, git_sha=GIT_SHA),
           "per_kernel_launch": {k: per_kernel_bytes[k] / per_kernel_n[k] for k in per_kernel_bytes if k.startswith("hn_")},
           "per_kernel_step": {k: per_kernel_bytes[k] / PMC_STEPS for k in per_kernel_bytes if k.startswith("hn_")},
           "note": "FETCH_SIZE x 1024 x 2 (gfx950 half-count of wide coalesced reads) + WRITE_SIZE x 1024, separate "
                   "rocprofv3 --pmc passes over an eager 5-step bench run; every kernel of the step included"}
json.dump(traffic, open(os.path.join(dst, f"{tag}_traffic_config{cfg}.json"), "w"), indent=1)
print(open(os.path.join(dst, f"{tag}_pmc_per_kernel{sfx}.csv")).read())
print(json.dumps(traffic, indent=1))
