#!/bin/bash
# PMC passes of the nltenebular bench step and their summary -> gpurun_out/pmc_neb_<round>/{pmc_summary_nltenebular.txt,pmc_traffic_nltenebular.json}
T=${1:-r03}
R=$GRAFT_REPO_ROOT
cd $R && bash tools/pmc_collect.sh 10000000 neb_$T nltenebular
mkdir -p $R/gpurun_out/pmc_neb_out
cd $R && PMC_TRAFFIC_JSON=$R/gpurun_out/pmc_neb_out/pmc_traffic_nltenebular.json python3 tools/pmc_summary.py gpurun_out/pmc_neb_$T/pass* > $R/gpurun_out/pmc_neb_out/pmc_summary_nltenebular.txt
rm -rf $R/gpurun_out/pmc_neb_$T
tail -5 $R/gpurun_out/pmc_neb_out/pmc_summary_nltenebular.txt
