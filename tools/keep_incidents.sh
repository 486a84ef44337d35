#!/bin/bash
# Copies what a GPU-side run left under gpurun_out/incidents/ (faulthandler records of crashed test processes, tests/conftest.py;
# logs a GPU script saved there) into profiles/incidents/, which is tracked: gpurun_out/ is scratch that every call overwrites.
set -e
cd "$(dirname "$0")/.."
mkdir -p profiles/incidents
shopt -s nullglob
n=0
for f in gpurun_out/incidents/*; do
	cp -n "$f" profiles/incidents/ && n=$((n + 1))
done
echo "kept $n incident file(s) in profiles/incidents/"
