// api_internal.h — state of api.hip shared with the other translation units of libhipstr_hmm.so.
#pragma once
#include <hip/hip_runtime.h>
#include <string>

namespace hipstr {

struct ApiTables {                 // device copies of HostTables, owned by api.hip (hipstr_hmm_init)
  const double *int_log, *qual_correct, *qual_error, *m2m, *m2i;
  hipStream_t stream;
};
int api_device_tables(ApiTables* t);          // initialises device 0 on first use; 1 + hipstr_last_error() on failure
int api_fail(const std::string& message);     // records hipstr_last_error(); returns 1

}  // namespace hipstr
