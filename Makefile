# Top-level build: HIP library (gfx950), plain-C host driver, CPU oracle (test infrastructure).
HIPCC ?= /opt/rocm/bin/hipcc
CC ?= gcc
PKG = ray-tracing_amd
LIBDIR = $(PKG)/lib
# -fno-slp-vectorize: packed FP32 instructions (v_pk_mul_f32, v_pk_add_f32) that the SLP vectoriser forms out of the
# scalar per-lane arithmetic cost more than the two instructions they replace on gfx950 (matrix kind +3 %)
HIPFLAGS = --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fvisibility=hidden -Wall
CFLAGS = -std=gnu11 -O2 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wextra

HIP_SRCS = $(wildcard $(PKG)/csrc/*.hip)
HIP_HDRS = $(wildcard $(PKG)/csrc/*.h) $(wildcard $(PKG)/csrc/*.hpp) $(wildcard include/*.h)

all: $(LIBDIR)/librt_amd.so $(LIBDIR)/librt_host.so $(LIBDIR)/rt_render oracle

# one object per source file, each with the extra flags csrc/FLAGS lists for it
OBJDIR = $(LIBDIR)/obj
HIP_OBJS = $(patsubst $(PKG)/csrc/%.hip,$(OBJDIR)/%.o,$(HIP_SRCS))
$(OBJDIR)/%.o: $(PKG)/csrc/%.hip $(HIP_HDRS) $(PKG)/csrc/FLAGS
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(shell sed -n 's/^$*\.hip //p' $(PKG)/csrc/FLAGS) -c -o $@ $<

$(LIBDIR)/librt_amd.so: $(HIP_OBJS)
	$(HIPCC) --offload-arch=gfx950 -fPIC -shared -o $@ $(HIP_OBJS)

$(LIBDIR)/librt_host.so: $(PKG)/host/rt_host.c $(PKG)/host/rt_host.h include/rt_abi.h
	@mkdir -p $(LIBDIR)
	$(CC) $(CFLAGS) -shared -o $@ $(PKG)/host/rt_host.c -lm

$(LIBDIR)/rt_render: $(PKG)/host/rt_cli.c $(PKG)/host/rt_renderer.c $(PKG)/host/rt_renderer.h $(LIBDIR)/librt_amd.so $(LIBDIR)/librt_host.so
	$(CC) $(CFLAGS) -o $@ $(PKG)/host/rt_cli.c $(PKG)/host/rt_renderer.c -L$(LIBDIR) -lrt_host -lrt_amd -lm -Wl,-rpath,'$$ORIGIN'

oracle:
	$(MAKE) -C oracle

# instruction-rate and matrix-core probes behind the design (their outputs are under profiles/)
TOOLS = tools/valu_rate tools/mfma_coexec tools/mfma_f16_probe tools/mfma_split_probe tools/op_rate tools/div_probe \
        tools/filter_probe tools/mfma_hazard_probe tools/cvt_probe tools/sign_probe
tools: $(TOOLS)
tools/%: tools/%.hip $(HIP_HDRS)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Iinclude -o $@ $<

host: $(LIBDIR)/librt_host.so

clean:
	rm -f $(LIBDIR)/*.so $(LIBDIR)/rt_render $(OBJDIR)/*.o
	$(MAKE) -C oracle clean

.PHONY: tools all oracle host clean
