// Driver of tests/test_guard.py: vk_guard.h (vectorian_amd/csrc) with a fake stream.  A "copy in flight" is a pointer the fake
// stream still holds; draining the stream completes the copies (it WRITES through the pointers).  The buffers are vectors of a
// probe type whose destructor logs: the log shows whether a failing body's buffers outlive the drain.  Built with
// -fsanitize=address, a write into a buffer that is already gone also aborts the process.
#include "vk_guard.h"

#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

static std::vector<std::string> g_log;

struct Probe {
	int value = 0;
	bool live = true;
	~Probe() { if (live) g_log.push_back("buffer destroyed"); live = false; }
};

struct FakeStream {
	std::vector<Probe *> in_flight;
	void copy_async(Probe *dst) { in_flight.push_back(dst); }
	void synchronize() {
		for (Probe *p : in_flight) p->value = 42;   // the DMA lands
		g_log.push_back(in_flight.empty() ? "drained (idle)" : "drained");
		in_flight.clear();
	}
};

static void report(const char *name, int rc) {
	printf("%s rc=%d:", name, rc);
	for (const auto &s : g_log) printf(" [%s]", s.c_str());
	printf("\n");
	g_log.clear();
}

int main() {
	FakeStream st;
	// 1. a body that fails with a copy in flight: drained BEFORE its buffer dies
	int rc = vk_run_guarded([&](vk_host_keep &keep) {
		std::vector<Probe> &buf = keep.vec<Probe>(1);
		st.copy_async(buf.data());
		return 3;   // e.g. VK_ERR_HIP from the launch that followed the copy
	}, [&]() { st.synchronize(); }, [](const char *) { return 1; });
	report("fail_with_copy_in_flight", rc);
	// 2. success: the body synchronised itself; no drain by the guard
	rc = vk_run_guarded([&](vk_host_keep &keep) {
		std::vector<Probe> &buf = keep.vec<Probe>(1);
		st.copy_async(buf.data());
		st.synchronize();
		return 0;
	}, [&]() { st.synchronize(); }, [](const char *) { return 1; });
	report("success", rc);
	// 3. an abort between the passes of a batch (VK_ERR_ABORTED = 6): as a failure
	rc = vk_run_guarded([&](vk_host_keep &keep) {
		int *stack_like = keep.array<int>(160);
		stack_like[159] = 7;
		std::vector<Probe> &a = keep.vec<Probe>(2), &b = keep.vec<Probe>(1);
		st.copy_async(&a[1]); st.copy_async(&b[0]);
		return 6;
	}, [&]() { st.synchronize(); }, [](const char *) { return 1; });
	report("abort_between_passes", rc);
	// 4. a C++ exception inside the body: reported as a status, drained first
	rc = vk_run_guarded([&](vk_host_keep &keep) -> int {
		std::vector<Probe> &buf = keep.vec<Probe>(1);
		st.copy_async(buf.data());
		throw std::runtime_error("bad_alloc stand-in");
	}, [&]() { st.synchronize(); }, [](const char *what) { g_log.push_back(std::string("exception: ") + what); return 1; });
	report("exception", rc);
	// 5. addresses handed out by the keep stay put when more buffers are asked for
	rc = vk_run_guarded([&](vk_host_keep &keep) {
		std::vector<int> &first = keep.vec<int>(8, 5);
		const int *where = first.data();
		for (int i = 0; i < 100; i++) keep.vec<double>(1000);
		return (first.data() == where && first[7] == 5 && keep.size() == 101) ? 0 : 9;
	}, [&]() { st.synchronize(); }, [](const char *) { return 1; });
	report("stable_addresses", rc);
	return 0;
}
