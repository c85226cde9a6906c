// placeholder: implemented next
#include "common.h"
extern "C" int mdf_conv3d_fwd(const float*, const float*, const float*, const float*, const float*, float*, int, int, int,
                              int, int, int, int, int, int, void*) {
  return mdf::fail(MDF_EUNSUPPORTED, "mdf_conv3d_fwd not built yet");
}
extern "C" int64_t mdf_conv3d_packed_size(int, int) { return 0; }
extern "C" int mdf_conv3d_pack_weights(const float*, float*, int, int, int, void*) {
  return mdf::fail(MDF_EUNSUPPORTED, "mdf_conv3d_pack_weights not built yet");
}
extern "C" int mdf_prob_softmax_regress_fwd(const float*, const float*, const float*, int, float*, float*, int, int, int, int,
                                            int, void*) {
  return mdf::fail(MDF_EUNSUPPORTED, "mdf_prob_softmax_regress_fwd not built yet");
}
