#!/bin/bash
timeout 600 python tools/rs_bench.py 2>&1 | tail -30
