#!/bin/bash
# round 5: the whole GPU suite with EVERY context taken from the check build (DSM_LIBRARY=check) -- "green against both builds":
# the default run of the suite uses the product library and switches to the check build only where a test sets a cross-check switch
out=gpurun_out/r5u
mkdir -p $out
DSM_LIBRARY=check timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_check_build.log 2>&1; tail -5 $out/pytest_check_build.log | cut -c1-200
