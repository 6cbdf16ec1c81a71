#!/bin/bash
# round 5 call 49: capped decoder rounds as shipped (2..8 byte symbols 4, 8 bit Short 6): full GPU suite
cd /root/repo
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -8
