run() { echo "$@"; env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-d2h 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_step'])"; }
run F3DG_RENDER_DMA=1
run F3DG_RENDER_DMA=0
run F3DG_RENDER_LDS_PAD=512
run F3DG_RENDER_LDS_PAD=1536
run F3DG_RENDER_LDS_PAD=3072
run F3DG_RENDER_LDS_PAD=5120
run F3DG_RENDER_LDS_PAD=8192
