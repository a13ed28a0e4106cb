// Augmented-Lagrangian driver: internal declarations (the state structs live in bq_common.h next to bq_solver).
#pragma once
#include "bq_common.h"

constexpr int BQ_AL = 4;   // internal solver kind (created by bq_al_solver_create)
