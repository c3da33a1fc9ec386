cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_ldm.py -q -x -m gpu -k primitives 2>&1 | grep -E "AssertionError|assert " | head -5
