#!/bin/bash
# GPU-box helper: end-to-end CLI timing (files -> .zwz -> files) of the product binary vs the reference binary.
# usage: tools/e2e_gpu.sh <kind: random|text> <files> <bytes>
K=${1:-random}; N=${2:-2000}; B=${3:-262144}
W=/tmp/zwz_e2e_$$; mkdir -p $W/data
python3 - <<PY
import sys, os
sys.path.insert(0, "$GRAFT_REPO_ROOT/tests")
import corpus
os.makedirs("$W/data/src", exist_ok=True)
for i in range($N):
    d = corpus.random_bytes(7000 + i, $B) if "$K" == "random" else corpus.text_like(7000 + i, $B)
    open("$W/data/src/f%05d.bin" % i, "wb").write(d)
PY
M=$GRAFT_REPO_ROOT/parallel-data-compression-and-decompression_amd/main
R=$GRAFT_REPO_ROOT/oracle/_ref/main
echo "== product ($K, $N x $B B)"
$M compress $W/data/src $W/zwz | grep "Time Taken"
ZWZ_VERBOSE=1 $M compress $W/data/src $W/zwz2 2>&1 | grep -E "Time Taken|zwz:"
ZWZ_VERBOSE=1 $M decompress $W/zwz $W/back 2>&1 | grep -E "Time Taken|zwz: "
echo "== reference"
[ -n "$SKIP_REF" ] && { rm -rf $W; exit 0; }
$R compress $W/data/src $W/rzwz > $W/r1.log 2>&1; grep "Time Taken" $W/r1.log
$R decompress $W/rzwz $W/rback > $W/r2.log 2>&1; grep "Time Taken" $W/r2.log
cmp $W/zwz/compressed_0.zwz $W/rzwz/compressed_0.zwz && echo "shards identical"
diff -rq $W/back $W/rback > /dev/null && echo "decoded trees identical"
rm -rf $W
