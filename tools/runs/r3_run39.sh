#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 300 python tools/diag_host_ahead.py
