// disassembly aid: instantiates the shipping k_gemm256p<bf16, plain absmax, AM4, byte table> alone
#include "../../mps_bitsandbytes_amd/csrc/gemm256.h"
namespace mbnb {
template __global__ void k_gemm256p<bf16_t, false, 0, true, true>(const bf16_t *, Q4ProducerRT<bf16_t, false>::Params, const bf16_t *, void *, int, int64_t, int64_t, int64_t);
}
