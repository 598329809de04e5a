# A/B of library builds on the headline step: [MATH=f22] bash tools/ab_bench.sh <grep pattern> [lib ...]   (the in-tree library first; default arithmetic: exact f32)
pat=$1; shift
for lib in "" "$@"; do
  echo "== lib ${lib:-in-tree}"
  IDEE_HIP_LIB=${lib:+$PWD/$lib} python bench.py ${MATH:+--math $MATH} --breakdown --steps 15 --warmup 5 --no-cpu-baseline --no-exact-line --no-fwd-eval 2>&1 >/dev/null | grep "$pat\|wall"
done
