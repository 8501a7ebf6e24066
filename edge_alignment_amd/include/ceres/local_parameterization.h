// ceres/local_parameterization.h — the reference's ROS-flavour headers include it by name (include/EAResidue.h:19);
// LocalParameterization and QuaternionParameterization live in the facade's ceres.h.
#pragma once
#include "ceres.h"
