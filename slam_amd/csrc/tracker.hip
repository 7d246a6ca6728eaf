// Frame-level host logic (placeholder until the tracker lands).
#include "ctx.hpp"
namespace ktlvo {
struct TrackerState {};
void tracker_free(ktlvo_ctx *ctx) { delete ctx->tracker; ctx->tracker = nullptr; }
} // namespace ktlvo
