cd /root/repo; export TMPDIR=/tmp
FRP_LIB=$PWD/forces_resilient_planner_amd/lib_aprof.so python tests/tools/astar_prof.py 256
tools/ab_variants.sh fdef flicm ftrk flt
