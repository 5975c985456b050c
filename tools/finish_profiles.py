"""Third step of a round's evidence (after tools/collect_profiles.sh + make_profiles.py + tools/final_lines.sh):
copies gpurun_out/<tag>_final_*.json into profiles/ and refreshes <tag>_eval_bench.json's runs.  Where the opt-in
8-bit-stash mode was collected too (round 3; `FINAL_S8=1 tools/final_lines.sh`, frozen since round 4) its PMC / kernel
stats passes are condensed into <tag>_traffic_config2_s8.json, <tag>_kernel_stats_config2_s8.csv.
    python tools/finish_profiles.py r04"""
import collections, csv, glob, json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
from importlib import import_module
L = import_module("hypernerf_torch_amd._lib")
src = f"{R}/gpurun_out/prof_{tag}_c2s8"
have_s8 = os.path.isdir(src) and os.path.exists(f"{R}/gpurun_out/{tag}_final_config2_s8.json")
tot = None
if have_s8:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ("pmc_fetch", "pmc_write"):
        f = glob.glob(f"{src}/{sub}/**/*counter_collection.csv", recursive=True)[0]
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[(name, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    STEPS = 5
    tot, per = 0.0, collections.defaultdict(float)
    for (name, grid), c in agg.items():
        n = max(len(v) for v in c.values())
        fb = sum(c.get("FETCH_SIZE", [0])) / max(1, len(c.get("FETCH_SIZE", [0]))) * 1024 * 2
        wb = sum(c.get("WRITE_SIZE", [0])) / max(1, len(c.get("WRITE_SIZE", [0]))) * 1024
        tot += (fb + wb) * n / STEPS
        per[name.split("<")[0]] += (fb + wb) * n
    out = {"what": "HBM bytes per step of the OPT-IN 8-bit-stash mode (bench.py --precision bf16s8, config 2), measured like "
                   f"{tag}_traffic_config2.json: FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024, separate rocprofv3 --pmc passes over an "
                   "eager 5-step run (tools/final_lines.sh)",
           "dtype": "bf16s8", "build": L.build_id(), "bytes_per_step": tot,
           "per_kernel_step": {k: v / STEPS for k, v in sorted(per.items(), key=lambda kv: -kv[1]) if v / STEPS > 1e6},
           "bf16_mode_bytes_per_step": json.load(open(f"{R}/profiles/{tag}_traffic_config2.json"))["bytes_per_step"]}
    json.dump(out, open(f"{R}/profiles/{tag}_traffic_config2_s8.json", "w"), indent=1)
    shutil.copy(glob.glob(f"{src}/stats/**/*kernel_stats.csv", recursive=True)[0], f"{R}/profiles/{tag}_kernel_stats_config2_s8.csv")
names = [("config2", "bench_line.json"), ("config3", "bench_line_config3.json"), ("config5", "bench_line_config5.json"),
         ("config1", "bench_line_config1.json"), ("config2_fp32", "bench_line_config2_fp32.json"),
         ("config2_dp1", "bench_line_config2_one_rank_rccl.json"), ("config2_s8", "bench_line_config2_s8.json")]
names = [(a, b) for a, b in names if os.path.exists(f"{R}/gpurun_out/{tag}_final_{a}.json")]
for a, b in names:
    shutil.copy(f"{R}/gpurun_out/{tag}_final_{a}.json", f"{R}/profiles/{tag}_{b}")
ev = f"{R}/profiles/{tag}_eval_bench.json"
d = json.load(open(ev)) if os.path.exists(ev) else {
    "what": "tools/eval_bench.py: inference.render_image, one 378x504 image (190,512 rays), BASELINE config-2 model (64 coarse + "
            "128 fine-level points per ray), bf16, deterministic eval branch, one MI355X; three chunk sizes"}
d["runs"] = []
for l in open(f"{R}/gpurun_out/{tag}_final_eval.jsonl"):
    r = json.loads(l)
    d["runs"].append({k: (round(v, 5) if isinstance(v, float) and k != "forward_flops" else v) for k, v in r.items() if k not in ("what", "build")})
d["build"] = L.build_id()
json.dump(d, open(ev, "w"), indent=1)
for a, _ in names:
    r = json.loads(open(f"{R}/gpurun_out/{tag}_final_{a}.json").read().strip().splitlines()[-1])
    pk = r["roofline"]["per_kernel"]
    print(a, round(r["value"] / 1e6, 2), "M", round(r["ms_per_step"], 4), "ms", {k.replace("hn_", "").replace("_kernel", ""): round(v["ms_per_step"], 4) for k, v in pk.items()},
          "traffic", r["roofline"].get("traffic"), "mfma step", round(r.get("step_mfma_frac", 0), 3), r["build"]["kernel_src_sha256"],
          "also" if "also" in r else "")
print("s8 bytes/step", None if tot is None else round(tot / 1e9, 3), [(x["chunk"], x["s_per_image"], x["mfma_frac_of_2.5PF"]) for x in d["runs"]])
