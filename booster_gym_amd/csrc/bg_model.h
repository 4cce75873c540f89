// Host-side model object behind the opaque `bg_model` of the C ABI: the flat numeric description the kernels consume plus the names the
// asset queries of the reference return (envs/t1.py:57,85-108).  Filled by bg_model_create (names empty) or bg_model_load_urdf.
#pragma once
#include <string>
#include <vector>

#include "../../include/booster_gym_amd.h"

struct bg_model {
    bg_model_desc desc;
    std::vector<std::string> body_names, dof_names;
};

int bg_set_error(int code, const char* msg);
int bg_model_validate(const bg_model_desc* d);  // topology / range checks shared by bg_model_create and bg_model_load_urdf (0 or a negative code)
