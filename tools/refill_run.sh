set -o pipefail
timeout -k 10 300 python3 tools/refill_check.py 2,16 131072 70001 || exit 1
for e in "STG_REFILL=0" "STG_REFILL=2,8" "STG_REFILL=2,16" "STG_REFILL=2,32" "STG_REFILL=4,16"; do
  ENVV="$e" LIBS="build/lib_new.so" bash tools/ab_sizes.sh 1 "131072 1" "262144 1" "524288 1" "262144 0"
done
