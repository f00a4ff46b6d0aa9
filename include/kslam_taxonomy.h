/*
 * kslam_taxonomy.h -- C ABI of the taxonomy stage after the host tail (the
 * "per-read LCA" part of SURVEY.md section 8f row N1).  Same library as
 * kslam.h; host-only.
 *
 * Replaces, in the reference (citations into /root/reference/):
 *
 *   TaxonomyDB(file) / readTaxonomyIndex     src/TaxonomyDatabase.h:87-93, 166-183
 *   getLowestCommonAncestor / getParentTaxID src/TaxonomyDatabase.h:185-231
 *   getScientificName / getRank / getLineage src/TaxonomyDatabase.h:233-265
 *   getTaxIDAtRank / isBelowInTree / isSubSpecies
 *                                            src/TaxonomyDatabase.h:306-349
 *   getResultFromPairedOverlaps (taxonomy id + read name; the gene list is not
 *   carried)                                 src/MetagenomicResults.h:88-112
 *   convertAlignmentsToIdentifiedTaxonomies_parallel
 *                                            src/MetagenomicResults.h:182-197
 *   writePerReadResults                      src/MetagenomicResults.h:455-463
 *   combineTaxonomies + sortResults + writeAbbreviatedResultsFile (read counts
 *   per taxon)                               src/MetagenomicResults.h:149-177, 237-274
 *
 *   the gene list of getResultFromPairedOverlaps (:88-112), combineTaxonomies /
 *   combineRangeOfIdentifiedTaxonomy with their genes and reads (:118-177),
 *   sortResults (:254-273), correctXML / getXML / writeResults (:213-224, 275-366)
 *   -- the XML report: kslam_taxreport_* below
 *
 * The reference walks hash-map lookups up the tree and compares root-ward paths
 * level by level; here the tree is a dense parent/depth array and the LCA of a
 * set is a fold of pairwise depth-aligned walks -- same answers, including the
 * reference's conventions: a path stops below the root (a parent id of 1 ends
 * it), so taxa under different top-level nodes have LCA 0; an id of 0, or ids
 * the tree does not know next to other ids, give 0.
 */
#ifndef KSLAM_TAXONOMY_H_
#define KSLAM_TAXONOMY_H_
#include "kslam_tail.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kslam_taxdb kslam_taxdb;

/* The <db>/taxDB text: four lines per node -- taxonomy id, parent id, scientific
 * name, rank (writeTaxonomyIndex, src/TaxonomyDatabase.h:153-165).  A repeated id
 * keeps its first record, as the reference's map insert does.  Errors (message
 * in kslam_tail_last_error()): a line count that is not a multiple of four or an
 * id line that is not a number (the reference throws from std::stoi), a cycle in
 * the parent links (the reference would not terminate). */
kslam_status kslam_taxdb_parse(const char *text, uint64_t len, kslam_taxdb **out);
void kslam_taxdb_free(kslam_taxdb *db);
uint64_t kslam_taxdb_size(const kslam_taxdb *db);

uint32_t kslam_taxdb_lca(const kslam_taxdb *db, const uint32_t *tax_ids, uint64_t n);
uint32_t kslam_taxdb_parent(const kslam_taxdb *db, uint32_t tax_id);
uint32_t kslam_taxdb_at_rank(const kslam_taxdb *db, uint32_t tax_id, const char *rank);
int32_t kslam_taxdb_is_below(const kslam_taxdb *db, uint32_t upper, uint32_t lower);
int32_t kslam_taxdb_is_subspecies(const kslam_taxdb *db, uint32_t tax_id);
/* text is malloc'ed (kslam_free); which: 0 scientific name, 1 rank, 2 lineage */
kslam_status kslam_taxdb_text(const kslam_taxdb *db, uint32_t tax_id, int which, char **text,
                              uint64_t *text_len);

/* The tree in the dense form the LCA walks (for a copy on the device, include/kslam_samtext.h): nodes numbered in file
 * order; up[n] = node of the parent the reference's getParentTaxID gives (0xFFFFFFFF: none), depth[n] = nodes on the
 * path up to the top-level node, node_tax[n] = its taxonomy id.  The arrays live as long as the tree.
 * kslam_taxdb_node: the node of a taxonomy id, 0xFFFFFFFF when the tree does not know it. */
kslam_status kslam_taxdb_dense(const kslam_taxdb *db, uint64_t *n_nodes, const uint32_t **up, const uint32_t **depth,
                               const uint32_t **node_tax);
uint32_t kslam_taxdb_node(const kslam_taxdb *db, uint32_t tax_id);

/* One batch: the taxonomy id of every read pair that kslam_tail_pairs returned
 * (LCA over the entries of its alignment pairs) into tax_ids[n_read_pairs], and
 * the per-read lines "identifier \t taxonomy id \n" the reference writes to
 * <out>_PerRead (malloc'ed, kslam_free; may be NULL to skip). */
kslam_status kslam_tail_classify(const kslam_tail_params *params, const kslam_reads_view *reads,
                                 const kslam_index_view *index, const kslam_taxdb *db,
                                 const kslam_read_pair *read_pairs, uint64_t n_read_pairs,
                                 const kslam_paired_overlap *pairs, uint64_t n_pairs,
                                 uint32_t *tax_ids, char **per_read_text, uint64_t *per_read_len);

/* End of run: the taxonomy ids of all classified read pairs, in order, to the
 * "<name> \t <percent of num_reads>" lines of <out>_abbreviated.  Keeps the
 * reference's bookkeeping quirk (src/MetagenomicResults.h:159-175): when no read
 * pair is unclassified (id 0), the first record of the lowest id is dropped.
 * The reference orders equal ids with an unstable parallel sort; input order is
 * kept here (it only matters for which record that quirk drops). */
kslam_status kslam_taxonomy_summary(const kslam_taxdb *db, const uint32_t *tax_ids, uint64_t n,
                                    uint64_t num_reads, char **text, uint64_t *text_len);

/* ---- the XML report (writeResults, src/MetagenomicResults.h:213-224) -------------------------
 * A report object collects one IdentifiedTaxonomy per read pair over the batches of a run
 * (src/SLAM.h:244-248) and renders the report at the end (src/SLAM.h:257-265).  The gene fields the
 * report prints beyond kslam_index_view come from the database columns (include/kslam_db.h); every
 * pointer may be NULL (empty strings / GeneID 0).  `index` must be the same index in every call.
 *
 * As in the reference: per read pair the best-overlapping gene (GenbankEntry::getGene,
 * src/GenbankTools.h:170-185) of every alignment pair, std::sort by geneSort + std::unique
 * (operator==: the surviving representative is the one libstdc++'s sort puts first); per taxon the
 * genes of all its read pairs, sorted and merged with counts, reads sorted by name; taxa by
 * (reads descending, id ascending); abundance = reads * 100.0 / num_reads printed with
 * std::to_string (six decimals).  combineTaxonomies' bookkeeping quirk is kept (without an
 * unclassified read pair the record at the front of the sorted vector is dropped from its group, :159-175).
 * The reference orders records of equal taxonomy id with an unstable PARALLEL sort, so which record that is --
 * and which of several equal genes represents its class -- depends on the thread count there.  Here the
 * records keep their input order, except that the dropped record is the one std::sort leaves in front: what
 * the reference writes with one thread (and with any number below 1 000 read pairs), the form the golden
 * files of tests/golden/ are recorded in. */
typedef struct kslam_taxreport kslam_taxreport;
typedef struct {
  const char *gene_locus_tag;      /* Gene::locusTag, by gene */
  const uint64_t *gene_locus_tag_off;
  const char *gene_reference;      /* Gene::referenceSequence */
  const uint64_t *gene_reference_off;
  const uint32_t *gene_id;         /* Gene::geneID */
} kslam_gene_extras;

kslam_status kslam_taxreport_create(kslam_taxreport **out);
void kslam_taxreport_free(kslam_taxreport *report);
/* one batch: tax_ids as returned by kslam_tail_classify for the same read pairs */
kslam_status kslam_taxreport_add_batch(kslam_taxreport *report, const kslam_reads_view *reads,
                                       const kslam_index_view *index,
                                       const kslam_read_pair *read_pairs, uint64_t n_read_pairs,
                                       const kslam_paired_overlap *pairs, uint64_t n_pairs,
                                       const uint32_t *tax_ids);
/* end of run: the text writeResults streams to the output file (malloc'ed, kslam_free) */
kslam_status kslam_taxreport_xml(const kslam_taxreport *report, const kslam_index_view *index,
                                 const kslam_gene_extras *extras, const kslam_taxdb *db,
                                 uint64_t num_reads, char **text, uint64_t *text_len);

#ifdef __cplusplus
}
#endif
#endif /* KSLAM_TAXONOMY_H_ */
