#!/bin/bash
# user-closure workload in the bench's plain back-to-back passes under HIP / ROCr settings that the tracer may change
# (profiles/r03/user_workload_plain_vs_traced.txt: 3.9 ms plain, 2.98 ms per dispatch under rocprofv3)
run() { name=$1; shift; env "$@" python bench.py --workload user --no-cpu-baseline --place-gib 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['roofline']['kernel_ms'],3))"; }
run default PMX_X=1
run dev_kernarg_1 HIP_FORCE_DEV_KERNARG=1
run dev_kernarg_0 HIP_FORCE_DEV_KERNARG=0
run serialize AMD_SERIALIZE_KERNEL=3
run one_queue GPU_MAX_HW_QUEUES=1
run no_async_reclaim HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0
run scratch_limit HSA_SCRATCH_SINGLE_LIMIT=2000000000
run default_again PMX_X=1
