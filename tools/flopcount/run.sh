#!/bin/sh
# exact algorithmic flop count of one env-step from the instrumented oracle (see flopcount.cpp)
cd "$(dirname "$0")" && g++ -O1 -std=c++17 -w -o /tmp/irrl_flopcount flopcount.cpp && for s in 1 3 0 2; do IRRL_FLOPCOUNT_SOLVER=$s /tmp/irrl_flopcount; done && /tmp/irrl_flopcount train
