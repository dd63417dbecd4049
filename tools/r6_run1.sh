set -x
mkdir -p gpurun_out/r6a
python tools/ab2.py 8192 base 17=2 17=4 > gpurun_out/r6a/ab_sub_8192.txt 2>&1
AB_LA=1 python tools/ab2.py 4096 base 17=2 17=4 > gpurun_out/r6a/ab_sub_4096.txt 2>&1
python tools/ab2.py 1500 base 17=2 17=4 > gpurun_out/r6a/ab_sub_1500.txt 2>&1
CUGP_TUNE="17=4" timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/r6a/gputest_s4.txt 2>&1
tail -3 gpurun_out/r6a/gputest_s4.txt
CUGP_TUNE="17=2" timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/r6a/gputest_s2.txt 2>&1
tail -3 gpurun_out/r6a/gputest_s2.txt
cat gpurun_out/r6a/ab_sub_*.txt
