"""MI355X-native hot path of mini-sglang: paged decode/extend attention over the slot pool,
fused RMSNorm / SiluAndMul / RoPE, behind the reference's attention-backend interface.

Layout:
  csrc/            HIP kernels (gfx950) + the C ABI declared in include/msgl_hip.h
  _capi.py         ctypes binding of libmsgl_hip.so (fails loudly when the library is missing)
  ops.py           tensor-level wrappers around the C ABI
  layers/          RMSNorm, SiluAndMul, RotaryEmbedding, Attention (reference module signatures)
  attn/            AttentionBackend interface + HipAttnBackend (``--attention_backend hip``)
  mem_cache/       ReqToTokenPool, MHAKVPool, PageAllocator (the tensors the kernels read)
  runtime/         the caller side needed to drive the path: BatchInfo, a Qwen3-shaped module
                   stack, ModelRunner, hipGraph decode runner, synthetic workloads
"""
__version__ = "0.1.0"
