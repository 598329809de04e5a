# A/B of library builds on the headline step: bash tools/ab_bench.sh <grep pattern> [lib ...]   (the in-tree library first)
pat=$1; shift
for lib in "" "$@"; do
  echo "== lib ${lib:-in-tree}"
  IDEE_HIP_LIB=${lib:+$PWD/$lib} python bench.py --breakdown --steps 15 --warmup 5 --no-cpu-baseline --no-exact-line --no-fwd-eval 2>&1 >/dev/null | grep "$pat\|wall"
done
