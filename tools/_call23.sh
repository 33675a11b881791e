cd $GRAFT_REPO_ROOT
run() {
  env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-contract-leg > gpurun_out/sw.json 2> gpurun_out/sw.err || tail -3 gpurun_out/sw.err
  python - "$*" <<'PY'
import json, sys
d=json.load(open("gpurun_out/sw.json"))
print(sys.argv[1], "ms/step %.2f" % d["ms_per_step"], {k: round(v,2) for k,v in {a: b["ms_per_step"] for a, b in d["roofline"]["kernel_utilisation"].items()}.items()})
PY
}
run X=0
for r in 4 8 16 64; do run CASTRO_AMD_FOLD_TILE_ROWS=$r; done
for r in 8 16 64; do run CASTRO_AMD_TILE_ROWS=$r CASTRO_AMD_FOLD_TILE_ROWS=32; done
for r in 8 32; do run CASTRO_AMD_FUSED_TILE_ROWS=$r; done
run X=0
