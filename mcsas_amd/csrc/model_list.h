// model_list.h — the built-in models by id (include/mcsas_hip.h: MCSAS_MODEL_*).  The ONE list the kernel translation
// units (Makefile: one kern_*.hip object per entry), the kernel lookups and the host's trait table (mcsas_hip.hip) are
// generated from.  A ninth built-in model is: `MCSAS_MODEL_X = 8` in include/mcsas_hip.h, `template <> struct
// Contrib<MCSAS_MODEL_X>` in models.h, `X(8)` here.  (Models that should need no rebuild: mcsas_hip_plugin_compile.)
#pragma once
#define MCSAS_FOR_MODELS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
