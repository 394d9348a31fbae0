#!/bin/bash
# the diagnostic build (lib_diag, -DTSP_STAMPS): never shipped, never timed as the product
R=${GRAFT_REPO_ROOT:-/root/repo}
make -C $R/tsp_optimization_amd/csrc OUT=../lib_diag -j8 HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall -Wno-unused-function -DTSP_STAMPS" > /dev/null && make -C $R/tsp_optimization_amd/host OUT=../lib_diag > /dev/null && echo diag build ok
