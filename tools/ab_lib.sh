# same-box A/B of two builds of the library: ddsp_piano_amd/libddspp_base.so (DDSPP_LIB) against the in-tree one
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for c in ${1:-headline}; do
    echo -n "BASE $c "; DDSPP_LIB=$GRAFT_REPO_ROOT/ddsp_piano_amd/libddspp_base.so python tools/bank_time.py $c 20 2>&1 | tail -1
    echo -n "NEW  $c "; python tools/bank_time.py $c 20 2>&1 | tail -1
  done
done
for i in 1 2; do
  echo -n "BASE step "; DDSPP_LIB=$GRAFT_REPO_ROOT/ddsp_piano_amd/libddspp_base.so python tools/trace_case.py headline dict 20 | tail -1
  echo -n "NEW  step "; python tools/trace_case.py headline dict 20 | tail -1
done
