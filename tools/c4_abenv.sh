#!/bin/bash
# Development tool (GPU box): A/B of environment-level variants of the C4 sample pass (tools/ab_c2.py c4)
#   usage: c4_abenv.sh "ENV=.. ENV=.. [FLAGS=-D..]" ...
for e in "$@"; do
  flags=""; envs=""
  for kv in $e; do case $kv in FLAGS=*) flags="${kv#FLAGS=}"; flags="${flags//@/ }";; *) envs="$envs $kv";; esac; done
  echo "env$envs flags $flags"; env $envs python tools/ab_c2.py ${AB_WHICH:-c4} "$flags" 2>&1 | tail -1 | cut -c1-170
done
