#!/bin/bash
# full GPU test suite + passbench of the product build: tools/gpu_test_pb.sh <tag> "<cfg1>;<cfg2>;..."
set -o pipefail
TAG=$1; CFGS=$2
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/$TAG/status.txt )
tail -c 1500 gpurun_out/$TAG/tests.log
bash tools/gpu_pb.sh $TAG "prod" "$CFGS"
