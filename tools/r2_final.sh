#!/bin/bash
# end-of-round validation: the whole GPU suite, the smoke entry, then the profile round
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/profile_round.sh r02 2>&1 | tail -25
