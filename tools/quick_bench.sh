#!/bin/bash
# usage: b.sh tag [bench args...]   -> prints value ms median
tag=$1; shift
python bench.py --no-cpu-baseline --no-roofline --no-extra "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'], d['ms_per_step_median'], d.get('loss_after'), d.get('c_abi_calls_per_step'))"
