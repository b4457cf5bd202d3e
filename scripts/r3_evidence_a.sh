# round 3 evidence, part A (after the code freeze): GPU suite, rocprofv3 kernel stats + counter passes for cfg3 and cfg5, the bound counters
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ea; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
tail -6 $O/pytest.txt
bash scripts/profile_round.sh r3 cfg3 > $O/profile_cfg3.txt 2>&1; tail -c 600 $O/profile_cfg3.txt; echo
bash scripts/profile_round.sh r3 cfg5 > $O/profile_cfg5.txt 2>&1; tail -c 400 $O/profile_cfg5.txt; echo
bash scripts/pmc_bound.sh $O/pmc_bound_cfg3 64 cfg3 > $O/pmc_bound_cfg3.txt 2>&1; grep -c "launches" $O/pmc_bound_cfg3.txt; grep "failed" $O/pmc_bound_cfg3.txt | head
