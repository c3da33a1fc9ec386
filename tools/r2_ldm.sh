cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_ldm.py -q -x -m gpu -s 2>&1 | grep -E "passed|failed|LDM 128|assert|Error" | tail -6
