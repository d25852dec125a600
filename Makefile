# Build the gfx950 engine (libpymoc_hip.so) and the CPU oracle (test infrastructure).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  = pymoc_amd/csrc
HIPHDR = $(wildcard $(CSRC)/*.h) include/pymoc_hip.h

all: lib oracle

HIPCFLAGS = --offload-arch=$(ARCH) -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-value $(EXTRA)
HIPOBJ = $(CSRC)/build/pymoc_hip.o $(CSRC)/build/column_g16.o $(CSRC)/build/column_g32.o $(CSRC)/build/column_g64.o $(CSRC)/build/equi.o $(CSRC)/build/equi_column.o $(CSRC)/build/jn2018_fast.o

lib: pymoc_amd/libpymoc_hip.so
$(CSRC)/build/%.o: $(CSRC)/%.hip $(HIPHDR)
	@mkdir -p $(CSRC)/build
	$(HIPCC) $(HIPCFLAGS) -c -o $@ $<
pymoc_amd/libpymoc_hip.so: $(HIPOBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(HIPOBJ) -ldl

oracle: oracle/libpymoc_oracle.so
oracle/libpymoc_oracle.so: oracle/pymoc_oracle.c oracle/pymoc_oracle.h
	gcc -O2 -ffp-contract=off -fPIC -shared -std=c99 -Wall -o $@ oracle/pymoc_oracle.c -lm

# host-side sanitizer target: the oracle built with ASan + UBSan; run the
# golden tests against it with
#   make oracle-asan && PYMOC_ORACLE_LIB=oracle/libpymoc_oracle_asan.so \
#     LD_PRELOAD=$$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 \
#     python -m pytest tests/test_oracle_golden.py -q
oracle-asan: oracle/libpymoc_oracle_asan.so
oracle/libpymoc_oracle_asan.so: oracle/pymoc_oracle.c oracle/pymoc_oracle.h
	gcc -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined \
	  -ffp-contract=off -fPIC -shared -std=c99 -Wall -o $@ oracle/pymoc_oracle.c -lm

clean:
	rm -rf pymoc_amd/libpymoc_hip.so oracle/libpymoc_oracle.so $(CSRC)/build

.PHONY: all lib oracle oracle-asan clean
