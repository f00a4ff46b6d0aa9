#!/bin/bash
# CPU-only sanitizer pass over the host stages (database load, FASTQ ingest, host tail): the GPU boxes cannot run
# sanitizers, these parts need no GPU.  Builds the two host benches with ASan+UBSan and with TSan and
# runs them on small inputs; any report fails the script.   usage: bash tools/sanitize_host.sh
set -e
cd "$(dirname "$0")/.."
T=$(mktemp -d)
FLAGS="-O1 -g -std=c++17 -pthread -fno-omit-frame-pointer"
g++ $FLAGS -fsanitize=address,undefined tools/tail_bench.cpp k-slam_amd/host/tail.cpp k-slam_amd/host/taxonomy.cpp -o $T/tail_asan
g++ $FLAGS -fsanitize=thread tools/tail_bench.cpp k-slam_amd/host/tail.cpp k-slam_amd/host/taxonomy.cpp -o $T/tail_tsan
g++ $FLAGS -fsanitize=address,undefined tools/fastq_bench.cpp k-slam_amd/host/fastq.cpp -o $T/fq_asan
g++ $FLAGS -fsanitize=thread tools/fastq_bench.cpp k-slam_amd/host/fastq.cpp -o $T/fq_tsan
g++ $FLAGS -fsanitize=address,undefined tools/db_check.cpp k-slam_amd/host/db.cpp k-slam_amd/host/tail.cpp -o $T/db_asan
g++ $FLAGS -fsanitize=thread tools/db_check.cpp k-slam_amd/host/db.cpp k-slam_amd/host/tail.cpp -o $T/db_tsan
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1 TSAN_OPTIONS=halt_on_error=1
for mode in 0 1 2; do $T/tail_asan 40000 6 2 $mode > /dev/null; $T/tail_tsan 20000 6 2 $mode > /dev/null; done   # mode 2: the batch loop's two-thread host stage
$T/fq_asan 40000 5 2 > /dev/null
$T/fq_tsan 40000 5 2 > /dev/null
$T/db_asan 150 $T > /dev/null
$T/db_tsan 150 $T > /dev/null
rm -rf $T
echo "sanitizers: clean (ASan+UBSan, TSan) on the host tail, the FASTQ ingest and the database load"
