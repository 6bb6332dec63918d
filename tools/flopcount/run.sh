#!/bin/sh
# exact algorithmic flop count of one env-step from the instrumented oracle (see flopcount.cpp)
cd "$(dirname "$0")" && g++ -O1 -std=c++17 -w -o /tmp/irrl_flopcount flopcount.cpp && /tmp/irrl_flopcount && /tmp/irrl_flopcount train
