// user_model.hpp -- a USER-SUPPLIED elementwise model (include/muse_model.h; the reference's SimpleMuseProblem closures,
// src/simple.jl:79-95, as compiled code).  When the library is built with -DMUSE_USER_MODEL_HEADER="<header>" the header's
// three functions are compiled for the device (models.hpp wraps them as UserModel<MAXB>, which the solver kernel is
// instantiated with exactly like a built-in model) and for the host (muse_engine.cpp checks the contract's zero-element
// requirements when a context is created).  Such a library holds ONLY the user model (model id MUSE_MODEL_USER).
#pragma once
#ifdef MUSE_USER_MODEL_HEADER
#include <math.h>
#if defined(__HIPCC__)
#define MUSE_MODEL_FN __host__ __device__ static inline __attribute__((always_inline))
#else
#define MUSE_MODEL_FN static inline
#endif
// where a launch's run-time constants sit in its kernel-argument block (args.hpp, BatchArgs::consts; muse_model.h, muse_const)
#include "args.hpp"
#define MUSE_KERNARG_CONSTS (muse::kArgsConstsOffset)
// what a header of the two-parameter family forms its coefficients with (include/muse_model.h, MUSE_MODEL_PAIR): the engine's exp --
// one fixed sequence of IEEE operations, the same on host and device (step.hpp) and in the CPU checker
#include "step.hpp"
#define muse_model_exp(x) muse::muse_exp(x)
#include MUSE_USER_MODEL_HEADER
#ifndef MUSE_MODEL_NAME
#error "the model header must #define MUSE_MODEL_NAME (include/muse_model.h)"
#endif
#endif
