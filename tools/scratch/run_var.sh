set -e
run() {
  make -C royaltracer-dx_amd -j8 2>&1 | grep -E "error" || true
  timeout 600 python -m pytest tests -m gpu -x -q -k "hostile or large_scene" 2>&1 | tail -1
  for w in sponza_1080p_16spp_8b bistro_1080p_16spp_8b; do
    timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $w 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['ms_per_step'])"
  done
}
echo "== A: rel 2.5e-5, pad 1e-5"
sed -i 's/kSlabLo = 0.99995f, kSlabHi = 1.00005f/kSlabLo = 0.999975f, kSlabHi = 1.000025f/' royaltracer-dx_amd/csrc/rtx_kernels.hip; run
echo "== B: rel 5e-5, pad 2e-6"
sed -i 's/kSlabLo = 0.999975f, kSlabHi = 1.000025f/kSlabLo = 0.99995f, kSlabHi = 1.00005f/' royaltracer-dx_amd/csrc/rtx_kernels.hip
sed -i 's/const float bvh_pad = 1e-5f \* scale;/const float bvh_pad = 2e-6f * scale;/' royaltracer-dx_amd/csrc/rtx_scene_host.cpp; run
echo "== C: rel 2.5e-5, pad 2e-6"
sed -i 's/kSlabLo = 0.99995f, kSlabHi = 1.00005f/kSlabLo = 0.999975f, kSlabHi = 1.000025f/' royaltracer-dx_amd/csrc/rtx_kernels.hip; run
