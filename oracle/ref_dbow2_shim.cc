// ref_dbow2_shim.cc -- TEST INFRASTRUCTURE ONLY.  C entry points over the reference's VENDORED DBoW2
// (/root/reference/Thirdparty/DBoW2: TemplatedVocabulary<FORB::TDescriptor, FORB>, exactly what include/ORBVocabulary.h
// typedefs as ORBVocabulary), compiled where it lies against oracle/ocv_shim.  Pins the oracle's / the device's
// TemplatedVocabulary::transform descent (orbo_bow_transform, orbx_bow_transform) against the reference code.
#include <string>
#include <vector>

#include "DBoW2/FORB.h"
#include "DBoW2/TemplatedVocabulary.h"

namespace {
struct Voc : public DBoW2::TemplatedVocabulary<DBoW2::FORB::TDescriptor, DBoW2::FORB> {
    using DBoW2::TemplatedVocabulary<DBoW2::FORB::TDescriptor, DBoW2::FORB>::transform;   // the per-feature overload is protected
    void one(const cv::Mat &f, int levelsup, int *word, int *node) const {
        DBoW2::WordId id; DBoW2::WordValue w; DBoW2::NodeId nid = 0;  // the reference leaves *nid unassigned for a leaf above the level asked for
        transform(f, id, w, &nid, levelsup);
        *word = (int)id; *node = (int)nid;
    }
};
}  // namespace

extern "C" {

void *dbowref_load_text(const char *path) {
    Voc *v = new Voc();
    if (!v->loadFromTextFile(path)) { delete v; return nullptr; }
    return v;
}
void dbowref_destroy(void *h) { delete static_cast<Voc *>(h); }
int dbowref_size(void *h) { return (int)static_cast<Voc *>(h)->size(); }

// per feature: word id and the node `levelsup` levels above the word (TemplatedVocabulary.h transform(feature, id, weight, nid, levelsup));
// plus the FeatureVector of the batch overload (Frame::ComputeBoW's call) flattened as (node id, feature index) pairs in map order
int dbowref_transform(void *h, const unsigned char *desc, int n, int levelsup, int *word, int *node, int *fv_node, int *fv_feat) {
    Voc *v = static_cast<Voc *>(h);
    std::vector<cv::Mat> feats(n);
    for (int i = 0; i < n; i++) {
        feats[i].create(1, 32, CV_8U);
        memcpy(feats[i].ptr(), desc + 32 * (size_t)i, 32);
        v->one(feats[i], levelsup, &word[i], &node[i]);
    }
    DBoW2::BowVector bv;
    DBoW2::FeatureVector fv;
    v->transform(feats, bv, fv, levelsup);
    int k = 0;
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it)
        for (size_t j = 0; j < it->second.size(); j++) { fv_node[k] = (int)it->first; fv_feat[k] = (int)it->second[j]; k++; }
    return k;
}

}  // extern "C"
