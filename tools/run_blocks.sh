set -e
make -s -C oracle
timeout -k 10 300 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for b in 256 128 64; do echo "== MD_BLOCK=$b"; MD_BLOCK=$b python tools/phase_profile.py 2>&1 | tail -8; done
MD_BLOCK=64 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
