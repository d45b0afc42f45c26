// Launch-group recorder (group.h).  State is per host thread; nothing here touches the device until egm_group_end().
#include "common.h"
#include "group.h"
#include <vector>
#include <stdlib.h>

namespace {
thread_local bool g_recording = false;
thread_local std::vector<EgmGroupRec> g_recs;
}  // namespace

bool egm_group_recording() { return g_recording; }
void egm_group_set_recording(bool on) { g_recording = on; }
void egm_group_push(const EgmGroupRec& r) { g_recs.push_back(r); }

extern "C" int egm_group_begin(void) {
    EGM_REQUIRE(!g_recording, "group_begin: a group is already open on this thread");
    g_recs.clear();
    g_recording = true;
    return EGM_OK;
}
extern "C" int egm_group_abort(void) {
    g_recs.clear();
    g_recording = false;
    return EGM_OK;
}
extern "C" int egm_group_end(egm_stream_t s) {
    EGM_REQUIRE(g_recording, "group_end: no open group");
    g_recording = false;
    std::vector<EgmGroupRec> recs;
    recs.swap(g_recs);
    // a member that fills the chip on its own gains nothing from sharing a launch (EGM_GROUP_MAX_GRID workgroups and up: alone)
    static const int max_grid = getenv("EGM_GROUP_MAX_GRID") ? atoi(getenv("EGM_GROUP_MAX_GRID")) : (1 << 30);
    std::vector<char> done(recs.size(), 0);
    for (size_t i = 0; i < recs.size(); ++i) {
        if (done[i]) continue;
        EgmGroupRec batch[EGM_GROUP_MAX];
        int n = 0;
        batch[n++] = recs[i]; done[i] = 1;
        for (size_t j = i + 1; j < recs.size() && n < EGM_GROUP_MAX && recs[i].grid < max_grid; ++j) {
            if (!done[j] && recs[j].launch == recs[i].launch && recs[j].grid < max_grid) { batch[n++] = recs[j]; done[j] = 1; }
        }
        const int rc = recs[i].launch(batch, n, (hipStream_t)s);
        if (rc != EGM_OK) return rc;
    }
    return EGM_OK;
}
