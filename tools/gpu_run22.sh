#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -4
