// mirge_native.hip -- host runtime + C ABI of libmirge_native.so (include/mirge_native.h).
//
// Host side of the MI355X hot path: device context with a stream-ordered buffer pool (no
// hipMalloc/hipFree once warm), library packing + k-mer table construction, and the launch
// sequences for pack / collapse / cascade / count join.  Kernels: mirge_kernels.hpp.
// There is no CPU implementation of any of the compute in this file or anywhere in the product.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/mirge_native.h"
#include "mirge_kernels.hpp"
#include "mirge_libbuild.hpp"

#ifndef MIRGE_BITMAP_MAXK
#define MIRGE_BITMAP_MAXK 10
#endif
static_assert(sizeof(mirge_policy) == sizeof(MirgePolicy), "policy layout");
static_assert(MIRGE_MAX_PASSES == MIRGE_MAX_PASSES_K, "pass cap");

#include "native_host.hpp"
#include "native_gz.hpp"
#include "native_ctx.hpp"
#include "native_lib.hpp"
#include "native_reads.hpp"
#include "native_collapse.hpp"
#include "native_cascade.hpp"
#include "native_join.hpp"
#include "native_csv.hpp"
#include "native_iso.hpp"
