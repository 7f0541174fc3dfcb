#!/bin/bash
# CPU only: the oracle built with UBSan and then with ASan, the whole -m "not gpu" suite on each (the sanitizers are not
# available on the GPU pool: this is the CPU half).  Restores the normal build afterwards.
cd "$(dirname "$0")/.."
cp oracle/liboracle.so /tmp/liboracle_keep.so
trap 'cp /tmp/liboracle_keep.so oracle/liboracle.so' EXIT      # the normal build comes back whatever happens
SRC="orc_fourier.c orc_encoder.c orc_decoder.c"
(cd oracle && gcc -O1 -g -fPIC -ffp-contract=off -fsanitize=undefined -shared -o liboracle.so $SRC -lm -lubsan)
python -m pytest tests -q -m "not gpu" -p no:cacheprovider > /tmp/ubsan.txt 2>&1; echo "UBSan: exit $?, reports: $(grep -c 'runtime error' /tmp/ubsan.txt); $(tail -1 /tmp/ubsan.txt)"
(cd oracle && gcc -O1 -g -fPIC -ffp-contract=off -fsanitize=address -shared -o liboracle.so $SRC -lm)
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so) python -m pytest tests -q -m "not gpu" -p no:cacheprovider > /tmp/asan.txt 2>&1; echo "ASan: exit $?, reports: $(grep -c 'ERROR: AddressSanitizer' /tmp/asan.txt); $(tail -1 /tmp/asan.txt)"
cp /tmp/liboracle_keep.so oracle/liboracle.so
