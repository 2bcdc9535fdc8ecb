#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 300 python tools/trace_small_launches.py 2>&1 | tail -40
