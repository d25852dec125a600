# Build the gfx950 engine (libpymoc_hip.so) and the CPU oracle (test infrastructure).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
HIPFLAGS = --offload-arch=$(ARCH) -O3 -ffp-contract=off -fPIC -shared -std=c++17 -Wno-unused-value
CSRC  = pymoc_amd/csrc
HIPSRC = $(CSRC)/pymoc_hip.hip
HIPHDR = $(wildcard $(CSRC)/*.h) include/pymoc_hip.h

all: lib oracle

lib: pymoc_amd/libpymoc_hip.so
pymoc_amd/libpymoc_hip.so: $(HIPSRC) $(HIPHDR)
	$(HIPCC) $(HIPFLAGS) -o $@ $(HIPSRC) -ldl

oracle: oracle/libpymoc_oracle.so
oracle/libpymoc_oracle.so: oracle/pymoc_oracle.c oracle/pymoc_oracle.h
	gcc -O2 -ffp-contract=off -fPIC -shared -std=c99 -Wall -o $@ oracle/pymoc_oracle.c -lm

clean:
	rm -f pymoc_amd/libpymoc_hip.so oracle/libpymoc_oracle.so

.PHONY: all lib oracle clean
