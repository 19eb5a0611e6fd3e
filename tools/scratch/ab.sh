for lib in libhipac_hip.so libhipac_bn256.so libhipac_bm256.so libhipac_bn256s3.so; do
  HIPAC_LIB_NAME=$lib python tools/opbench.py bf16 20 2>&1 | tail -1 | tr ' ' '\n' | grep -E "l3b0c1|l4b0c1|l2b0c1" | tr '\n' ' '; echo " <- $lib"
done
