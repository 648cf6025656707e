#!/bin/bash
# PMC passes only (part of tools/r02_evidence.sh): separate rocprofv3 --pmc runs, no trace domains.
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r02_evidence
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for grp in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  d=/tmp/pmc_$(echo $grp | tr ' ' '_')
  rm -rf $d
  rocprofv3 --pmc $grp --output-format csv -d $d -- python3 $R/tools/pmc_only.py $out/pmc_manifest.json > /dev/null 2> $out/pmc_last.err; echo "pmc [$grp] rc=$?"
done
cd $R
python3 tools/pmc_summary2.py $out/pmc_manifest.json $out/r02_pmc.json /tmp/pmc_* > /dev/null; echo "pmc summary rc=$?"
