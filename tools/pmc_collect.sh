#!/bin/bash
# HBM traffic of the update kernel from PMC counters, collected as MI355X_MICROARCH.md §HBM prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (no trace domains), gfx950 correction applied by
# tools/pmc_summary.py (FETCH_SIZE counts 64 B per 128-B request for 16-B/lane streaming reads: x2).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pmc_f /tmp/pmc_w
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -- python3 tools/k1_only.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -- python3 tools/k1_only.py > /dev/null 2>&1
python3 tools/pmc_summary.py /tmp/pmc_f /tmp/pmc_w gpurun_out/r01_k1_pmc.json
