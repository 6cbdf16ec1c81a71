#!/bin/bash
# round 5 call 42: full GPU suite on the build with the Short pp encoders, the async mono decode and the low-entropy helpers
cd /root/repo
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -8
