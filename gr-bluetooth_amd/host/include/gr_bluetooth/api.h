#pragma once
// symbol visibility for the block library (same macro name as the reference's api.h)
#if defined(__GNUC__)
#define GR_BLUETOOTH_API __attribute__((visibility("default")))
#else
#define GR_BLUETOOTH_API
#endif
