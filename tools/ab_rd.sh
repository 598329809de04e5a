# isolated timings of the f32-storage convs (A/B of libraries through IDEE_HIP_LIB): bash tools/ab_rd.sh [lib ...]
for lib in "" "$@"; do
  echo "== lib ${lib:-in-tree}"
  for c in "k333" "k133" "k233 16 16 --B 24 --wsets 6" "k233"; do
    IDEE_HIP_LIB=${lib:+$PWD/$lib} python tools/conv_bench_f32.py $c 2>&1 | grep -v amdgpu.ids
  done
done
