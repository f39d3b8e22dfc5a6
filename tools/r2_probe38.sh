#!/bin/bash
cd "$GRAFT_REPO_ROOT"; ./tools/probe_mfma
