# (GPU box) for every N: the case-205 repro with the unit compiled under -opt-bisect-limit=N (precompiled by tools/repro_compile.py): mismatching histogram entries
python -c "import torch" >/dev/null 2>&1
for N in "$@"; do
  r=$(MCI_JIT_FLAGS="-O3 -mllvm -opt-bisect-limit=$N" R_ONLY_LANES=8 R_NPB=200 R_NCHAIN=1 R_NBLK=1 timeout 120 python tools/repro_case.py 205 2>&1 | grep "^lanes" | sed -E 's/.*(hist mismatches [0-9]+).*/\1/')
  echo "N=$N -> $r"
done
