#!/bin/bash
# usage: tools/pmc_pass.sh <tag> "<counters>" -- <program> [args]   (run on the GPU box; one PMC pass per call)
tag=$1; ctrs=$2; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/pmc_$tag -- "$@" > $R/gpurun_out/pmc_$tag/run.log 2>&1
echo "pmc $tag exit=$?"
