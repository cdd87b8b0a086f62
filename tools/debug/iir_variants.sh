# tools/debug/iir_iq_time.py through every build/variants/lib_*.so and the product library, same call
for l in "" build/variants/lib_*.so; do echo "== ${l:-product}"; DD_LIB_PATH=$l python tools/debug/iir_iq_time.py 2>&1 | grep "c64 in place" | tail -2; done
