#!/bin/bash
# ThreadSanitizer run of the superblock scheduler's queue protocol (thor_amd/csrc/tk_sched.h) driven by OS threads
# (tests/hostsim/sched_stress.cpp).  Expected: "ok" lines only.
set -e
cd "$(dirname "$0")/.."
g++ -std=c++17 -O1 -g -fsanitize=thread -pthread -o /tmp/sched_stress_tsan tests/hostsim/sched_stress.cpp
TSAN_OPTIONS="halt_on_error=1" /tmp/sched_stress_tsan 16 17 30 12 1
TSAN_OPTIONS="halt_on_error=1" /tmp/sched_stress_tsan 40 5 6 24 9
TSAN_OPTIONS="halt_on_error=1" /tmp/sched_stress_tsan 64 3 3 16 5
