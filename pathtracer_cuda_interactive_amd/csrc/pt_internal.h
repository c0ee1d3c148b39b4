// pt_internal.h — shared by the translation units of libpt_hip.so (not part of the C ABI).
#pragma once
#include <string>

// Records the message pt_last_error() returns (thread-local) and hands `code` back.
int pt_fail(int code, const std::string& msg);
