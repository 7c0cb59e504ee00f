#!/bin/bash
# current build vs pharmsol_amd/lib/ab/base.so on several allocations each (tools/alloc_tune.py prints per allocation)
PYTHONPATH=$PWD python tools/alloc_tune.py 2>&1 | grep allocation | sed 's/^/current  /'
PMX_LIB=$PWD/pharmsol_amd/lib/ab/base.so PYTHONPATH=$PWD python tools/alloc_tune.py 2>&1 | grep allocation | sed 's/^/base     /'
