#!/bin/bash
bash scripts/cbow_wide_ab.sh 2>&1 | tail -20
