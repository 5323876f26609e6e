# controls kernel time against the persistent-grid size (workgroups per CU)
cd $GRAFT_REPO_ROOT
for w in 2 4 8 16 10000; do
  echo "DDSPP_CTL_WGS_PER_CU=$w"
  DDSPP_CTL_WGS_PER_CU=$w bash tools/trace1.sh headline ctl_$w 2>&1 | grep "ms per step\|inharmonic_controls"
done
