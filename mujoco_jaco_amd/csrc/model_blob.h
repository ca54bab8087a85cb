#pragma once
#include <cstddef>
#include <string>
#include <vector>

#include <jaco/model_dev.h>

// Fills *m and the float4-packed hull vertex table from a JACOMDL1 blob. Returns 0, or -1 with *error set.
int jaco_model_from_blob(const void* buf, size_t size, JacoModelDev* m, std::vector<float>* hull, std::string* error);
