#!/bin/bash
# Extends the shipped MIOpen user databases (ursabench_amd/miopen_db/) by the evaluation-mode shapes of the BMA leg at the batch
# sizes the twin merges to since round 3 (4,096 / 1,808 rows): MIOpen's own exhaustive search (MIOPEN_FIND_ENFORCE=3 tunes what the
# performance database lacks and leaves existing entries alone), started from a copy of the shipped files, recorded while ONLY
# that evaluation runs (tools/miopen_tune_bma.py). New lines are printed as a diff; copy the resulting files over the shipped ones.
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/miopen_tune_bma.sh'
R=$GRAFT_REPO_ROOT
db=$R/gpurun_out/miopen_db_bma
rm -rf $db; mkdir -p $db
cp $R/ursabench_amd/miopen_db/*.txt $db/
export MIOPEN_USER_DB_PATH=$db
export MIOPEN_FIND_ENFORCE=3
cd $R
T0=$(date +%s)
timeout -k 10 1000 python3 tools/miopen_tune_bma.py > $db.log 2> $db.err; echo "tune rc=$? t=$(( $(date +%s) - T0 ))s"
tail -2 $db.log
for f in $db/*.txt; do b=$(basename $f); echo "== $b: $(wc -l < $R/ursabench_amd/miopen_db/$b) -> $(wc -l < $f) lines"; done
diff <(sort $R/ursabench_amd/miopen_db/*ufdb.txt) <(sort $db/*ufdb.txt) | cut -c1-140 | head -60
