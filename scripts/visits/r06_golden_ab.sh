#!/bin/bash
export TMPDIR=/tmp
python scripts/debug/golden_grad_errors.py 2>&1 | grep -v "Warning\|detach\|print(" | tail -16
