/* fdw_config.h -- key=value deck reader shared by stencil_code and rtm_code (input.dat format of the
 * reference: cuda_reference_stencil_computation/input.dat, cuda_reference_RTM/models/<m>/input.dat). */
#ifndef FDW_CONFIG_H
#define FDW_CONFIG_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct fdw_deck fdw_deck;

/* Reads `path`; returns NULL (and prints to stderr) if it cannot be opened -- the reference exits with
 * EXIT_FAILURE there (functions.c:51-52, fd-source-code.cu:40-41). */
fdw_deck *fdw_deck_read(const char *path);
void fdw_deck_free(fdw_deck *d);

/* Lookups by EXACT key (the reference matches the first line that merely CONTAINS the key,
 * functions.c:55; on every deck it ships the two rules pick the same line for every key it uses
 * except the unused `rnd`).  Missing key: -1 / -1.0f / NULL, exactly the reference's sentinels
 * (functions.c:64,86,109).  Numeric conversion is atoi / atof like the reference. */
int fdw_deck_int(const fdw_deck *d, const char *key);
float fdw_deck_float(const fdw_deck *d, const char *key);
const char *fdw_deck_str(const fdw_deck *d, const char *key);
int fdw_deck_has(const fdw_deck *d, const char *key);

#ifdef __cplusplus
}
#endif
#endif
