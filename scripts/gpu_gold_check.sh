#!/bin/bash
# Runs the product CLI on the reference's test inputs on the GPU box and diffs the SAM against gold (@PG normalised).
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/gold_check
mkdir -p $OUT
make -C oracle -s
oracle/_build/oracle_build_filter -b 64 -k 19 -nh 2 --bits 67108864 -o $OUT/small.filter --split tests/golden/input/64-viral-genomes.fa
for mode in "t1:" "tag.t1:-s 10"; do
  n=${mode%%:*}; o=${mode#*:}
  timeout -k 10 120 dream_yara_amd/csrc/dream_yara_mapper -fi $OUT/small.filter tests/golden/gold/64-viral-binned-indices/ tests/golden/input/64-viral-reads.fa -o $OUT/out.$n.sam -e 3 --threads 1 $o -v 2> $OUT/log.$n.txt
  cat $OUT/log.$n.txt
  grep -v '^@PG' $OUT/out.$n.sam > $OUT/a.sam
  grep -v '^@PG' tests/golden/gold/64-viral-reads.$n.sam > $OUT/b.sam
  if diff $OUT/a.sam $OUT/b.sam > $OUT/diff.$n.txt; then echo "GOLD $n: IDENTICAL"; else echo "GOLD $n: $(grep -c '^<' $OUT/diff.$n.txt) records differ"; head -20 $OUT/diff.$n.txt | cut -c1-250; fi
done
