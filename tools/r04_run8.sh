cd /root/repo; export TMPDIR=/tmp
tools/ab_variants.sh noinfo
