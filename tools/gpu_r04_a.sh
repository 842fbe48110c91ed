#!/bin/bash
# r04 call A: new-path correctness, phased-schedule A/B, slack stagger A/B, PMC of the residual kernel, full GPU suite
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_phased.py tests/test_gpu_grouped_gemm.py -x -q > $O/r04a_tests_new.txt 2>&1; tail -5 $O/r04a_tests_new.txt
timeout 600 python tools/step_ab.py --arms base,phases2+lead2,phases2+lead4,phases2+lead6,phases2+lead9,phases2+lead13,phases2+lead4+gm8 --rounds 2 --steps 20 --no-graph > $O/r04a_phases_ab.txt 2>&1; tail -16 $O/r04a_phases_ab.txt
timeout 600 python tools/step_ab.py --arms base,slack50,slack100,stagger0 --rounds 2 --steps 20 --no-graph > $O/r04a_slack_ab.txt 2>&1; tail -10 $O/r04a_slack_ab.txt
timeout 900 bash tools/gpu_pmc_residual.sh > $O/r04a_pmc_residual.txt 2>&1; tail -4 $O/r04a_pmc_residual.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/r04a_gpu_suite.txt 2>&1; tail -5 $O/r04a_gpu_suite.txt
