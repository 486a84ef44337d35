// vk_guard.h -- no early return with copies in flight.
//
// vk_query / vk_query_batch hand function-local host buffers to hipMemcpyAsync (pageable memory: the runtime may pin the pages and
// let the DMA engine read or write them after the call has returned) and synchronise the handle's stream before they use or drop
// them.  An error between the copy and that synchronisation -- a failed launch, a failed allocation, the caller's abort flag --
// used to `return` at once: the buffers died with the stack frame while the copy could still be in flight (undefined behaviour;
// the reference unwinds RAII state through exceptions, vectorian/core/cpp/query.cpp:10-30).
//
// Now every such buffer lives in a `vk_host_keep` owned by the ENTRY POINT's wrapper, and the wrapper drains the stream before the
// keep dies whenever the body fails:
//
//     int vk_query(...) { return vk_run_guarded([&](vk_host_keep &keep) { return query_body(..., keep); }, drain_stream); }
//
// Host only, no HIP types: tests/test_guard.py compiles it with g++ and drives it with a fake stream (CPU tier).
#ifndef VK_GUARD_H
#define VK_GUARD_H

#include <cstddef>
#include <exception>
#include <memory>
#include <vector>

class vk_host_keep {
	struct slot { virtual ~slot() {} };
	template <typename T> struct vec_slot : slot { std::vector<T> v; };
	std::vector<std::unique_ptr<slot>> slots;

public:
	vk_host_keep() = default;
	vk_host_keep(const vk_host_keep &) = delete;
	vk_host_keep &operator=(const vk_host_keep &) = delete;

	// a vector that lives until the entry point returns (its address is stable: the vector object itself is heap-allocated)
	template <typename T> std::vector<T> &vec(size_t n = 0) {
		auto s = std::make_unique<vec_slot<T>>();
		s->v.resize(n);   // value-initialised
		std::vector<T> &ref = s->v;
		slots.push_back(std::move(s));
		return ref;
	}
	template <typename T> std::vector<T> &vec(size_t n, const T &fill) {
		std::vector<T> &ref = vec<T>(0);
		ref.assign(n, fill);
		return ref;
	}

	// n value-initialised elements (what a stack array used to be)
	template <typename T> T *array(size_t n) { return vec<T>(n).data(); }

	size_t size() const { return slots.size(); }
};

// Runs body(keep).  On failure (a status != 0, or a C++ exception, reported as `on_exception()`'s status) calls drain() -- the
// stream's synchronisation -- BEFORE the keep and its buffers are destroyed.  A successful body has synchronised already.
template <typename Body, typename Drain, typename OnException>
int vk_run_guarded(Body &&body, Drain &&drain, OnException &&on_exception) {
	vk_host_keep keep;
	int rc;
	try {
		rc = body(keep);
	} catch (const std::exception &e) {
		rc = on_exception(e.what());
	} catch (...) {
		rc = on_exception("unknown exception");
	}
	if (rc != 0) drain();
	return rc;   // `keep` dies here, after the drain
}

#endif
