for gb in 20 8 4 20 8 4; do echo "== budget $gb GiB"; TWOPACO_PART_BUDGET_GB=$gb timeout 300 python tools/e2e_cli.py m2 2>&1 | grep -E "round:|rounds|context|run "; done
