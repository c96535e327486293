// Feature switches of the render path: the names and defaults of the reference's src/config.h:3-8.
// There they select code at compile time; here they are read once, when a Renderer is constructed,
// and passed to the kernels as flags (ptk_set_flags), so one library build serves every setting.
#pragma once

#define PT_DEBUG_NORMAL            0
#define PT_RUSSIAN_ROULETTE        1
#define PT_DIRECT_LIGHT_SAMPLING   1
#define PT_INDIRECT_LIGHT_SAMPLING 1
#define PT_IMPORTANCE_SAMPLE       1
#define PT_CHECK_HEMISPHERE        1
