# same-box A/B of two library builds by kernel trace: ab_trace.sh <case> <kernel substring>
cd $GRAFT_REPO_ROOT
for v in BASE NEW BASE NEW; do
  if [ $v = BASE ]; then export DDSPP_LIB=$GRAFT_REPO_ROOT/ddsp_piano_amd/libddspp_base.so; else unset DDSPP_LIB; fi
  echo -n "$v "; bash tools/trace1.sh ${1:-headline} abt_$v 2>&1 | grep "${2:-inharmonic_controls}" | awk '{print $(NF-2), "us"}'
done
