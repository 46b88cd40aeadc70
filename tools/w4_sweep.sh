for v in "" DMA_STRIDE=6 DMA_STRIDE=5+DMA_FIRST=24 BAR_B=104+FA_FIRST=104+FA_STRIDE=1 BAR_A=24+DMA_FIRST=24+DMA_STRIDE=6 ""; do
  if [ -z "$v" ]; then unset ULL_LIB_PATH; echo "== shipped"; else export ULL_LIB_PATH=$PWD/tools/probes/lib_$v.so; echo "== $v"; fi
  SHAPES=llama FORMS=ship python tools/gemm_shapes.py 2>&1 | grep -v lm_head | tail -5 | cut -c1-95
done
