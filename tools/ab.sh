# same-box A/B of the working tree against the tree in ab_old/ (a git worktree of an earlier commit, built here)
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for form in dict audio; do
    (cd ab_old && python tools/trace_case.py ${1:-headline} $form 20 | tail -1 | sed "s/^/OLD /")
    python tools/trace_case.py ${1:-headline} $form 20 | tail -1 | sed "s/^/NEW /"
  done
done
