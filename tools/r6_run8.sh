O=gpurun_out/r6h
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gputest.txt 2>&1; tail -3 $O/gputest.txt
python tools/ab2.py 1500 base 12=0 > $O/ab_fuse_1500.txt 2>&1; cat $O/ab_fuse_1500.txt
python tools/ab2.py 4096 base 12=0 > $O/ab_fuse_4096.txt 2>&1; cat $O/ab_fuse_4096.txt
AB_LA=0 python tools/ab2.py 8192 base 12=100000 > $O/ab_fuse_8192.txt 2>&1; cat $O/ab_fuse_8192.txt
python tools/bcm_ab.py 12=2100 12=0 > $O/bcm_fuse.txt 2>&1; cat $O/bcm_fuse.txt
