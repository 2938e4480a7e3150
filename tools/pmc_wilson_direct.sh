# kept for the round-2 collection script: the PMC passes of tools/wilson_direct_bench.py
bash $(dirname $0)/pmc_ab.sh tools/wilson_direct_bench.py
