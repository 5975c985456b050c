export HN_PROF=1
python -c "import hypernerf_torch_amd._lib as L; L.build(force=True)"
for p in bf16 bf16s8; do echo "== $p"; HN_PRECISION=$p timeout 300 python tools/wg_prof.py 2>&1 | grep "^blk" | head -14; done
unset HN_PROF
python -c "import hypernerf_torch_amd._lib as L; L.build(force=True)"
