#!/bin/sh

python3 -m robosat_amd.tools "$@"
