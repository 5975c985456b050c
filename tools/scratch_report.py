"""Where do the machine kernels touch scratch?  Cross-compiles hn_mlp.hip to gfx950 assembly (no GPU needed) and, per
kernel, lists every basic block that holds a scratch_load / scratch_store next to its MFMA count: spills inside an
MFMA tile loop would show as blocks with both.   python tools/scratch_report.py > profiles/rNN_scratch_report.txt"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "hypernerf-torch_amd", "csrc", "hn_mlp.hip")
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "k.s")
    res = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                          "-munsafe-fp-atomics", "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                          "-o", out, src], capture_output=True, text=True)
    if res.returncode:
        sys.exit(res.stderr)
    asm = open(out).read().splitlines()
usage = {}
cur = None
for line in res.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = usage.setdefault(m.group(1), {})
    for key in ("VGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]"):
        m2 = re.search(r"remark:\s+" + re.escape(key) + r": (\d+)", line)
        if m2 and cur is not None:
            cur[key] = int(m2.group(1))
for name in ("_Z17hn_mlp_fwd_kernelILb1ELi2ELb0ELb1ELb0EEv9HnMlpArgs", "_Z17hn_mlp_fwd_kernelILb1ELi2ELb0ELb0ELb0EEv9HnMlpArgs",
             "_Z17hn_mlp_fwd_kernelILb1ELi2ELb1ELb1ELb0EEv9HnMlpArgs", "_Z17hn_mlp_bwd_kernelILb1ELb0ELb0EEv9HnMlpArgs", "_Z17hn_mlp_bwd_kernelILb1ELb1ELb0EEv9HnMlpArgs",
             "_Z15hn_wgrad_kernelILb1ELb0EEv14HnDwBatchTable",
             # the opt-in 8-bit-stash builds
             "_Z17hn_mlp_fwd_kernelILb1ELi2ELb0ELb1ELb1EEv9HnMlpArgs", "_Z17hn_mlp_bwd_kernelILb1ELb0ELb1EEv9HnMlpArgs",
             "_Z15hn_wgrad_kernelILb1ELb1EEv14HnDwBatchTable"):
    start = next(i for i, l in enumerate(asm) if l.startswith(name + ":"))
    end = next(i for i in range(start, len(asm)) if "s_endpgm" in asm[i])
    blocks, blk = [], ["entry", 0, 0, 0]
    blocks.append(blk)
    for l in asm[start:end]:
        m = re.match(r"(\.LBB\S+):", l)
        if m:
            blk = [m.group(1), 0, 0, 0]
            blocks.append(blk)
        blk[1] += "v_mfma" in l
        blk[2] += "scratch_load" in l
        blk[3] += "scratch_store" in l
    tot_m = sum(b[1] for b in blocks)
    print(f"{name}: {usage.get(name)}; {len(blocks)} basic blocks, {tot_m} MFMA instructions, "
          f"{sum(b[2] for b in blocks)} scratch loads, {sum(b[3] for b in blocks)} scratch stores")
    both = [b for b in blocks if b[1] and (b[2] or b[3])]
    for b in blocks:
        if b[2] or b[3]:
            print(f"    {b[0]:12s} mfma {b[1]:3d}  scratch_load {b[2]:3d}  scratch_store {b[3]:3d}")
    print(f"    blocks holding MFMAs AND scratch traffic: {len(both)} "
          f"({sum(b[1] for b in both)} of the {tot_m} MFMAs sit in them)")
