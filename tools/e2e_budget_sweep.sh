#!/bin/bash
# End-to-end CLI wall time against the partition-buffer budget, isolated (3 s between runs) and back to back: the driver wipes a
# finished process's device memory asynchronously and an allocation made meanwhile waits for it, so a run right behind another
# one pays for every byte it allocates.  Usage (GPU box, repo root): bash tools/e2e_budget_sweep.sh > gpurun_out/e2e_budget.txt
for b in 40 16 8 4; do
  sleep 8
  echo "== budget $b GiB, isolated (3 s pause)"
  TWOPACO_PART_BUDGET_GB=$b E2E_PAUSE=3 E2E_RUNS=3 python3 tools/e2e_cli.py m2 2>&1 | grep -E "^run |rounds \(|partition buffers|exec -> output complete"
  echo "== budget $b GiB, back to back"
  TWOPACO_PART_BUDGET_GB=$b E2E_PAUSE=0 E2E_RUNS=8 python3 tools/e2e_cli.py m2 2>&1 | grep -E "^run |rounds \(|partition buffers"
done
