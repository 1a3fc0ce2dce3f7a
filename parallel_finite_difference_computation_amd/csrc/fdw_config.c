/* fdw_config.c -- see fdw_config.h.  One pass over the file, any key order, '#' starts a comment,
 * surrounding blanks and the trailing CR/LF are dropped (the shipped decks have CRLF-free lines but the
 * DPC++ copies of the library are CRLF). */
#include "fdw_config.h"

#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct fdw_deck {
    int n, cap;
    char **key, **val;
};

static char *trim(char *s)
{
    while (*s && isspace((unsigned char)*s)) s++;
    char *e = s + strlen(s);
    while (e > s && isspace((unsigned char)e[-1])) *--e = '\0';
    return s;
}

fdw_deck *fdw_deck_read(const char *path)
{
    FILE *fp = fopen(path, "r");
    if (!fp) {
        fprintf(stderr, "cannot open input deck '%s'\n", path ? path : "(null)");
        return NULL;
    }
    fdw_deck *d = (fdw_deck *)calloc(1, sizeof *d);
    char *line = NULL;
    size_t len = 0;
    int oom = d == NULL;
    while (!oom && getline(&line, &len, fp) != -1) {
        char *hash = strchr(line, '#');
        if (hash) *hash = '\0';
        char *eq = strchr(line, '=');
        if (!eq) continue;
        *eq = '\0';
        char *k = trim(line), *v = trim(eq + 1);
        if (!*k) continue;
        if (d->n == d->cap) {      /* grow both tables or neither: a failed realloc leaves the old block (and the deck) intact */
            const int cap = d->cap ? 2 * d->cap : 32;
            char **nk = (char **)realloc(d->key, (size_t)cap * sizeof(char *));
            if (nk) d->key = nk;
            char **nv = nk ? (char **)realloc(d->val, (size_t)cap * sizeof(char *)) : NULL;
            if (nv) d->val = nv;
            if (!nk || !nv) {
                oom = 1;
                break;
            }
            d->cap = cap;
        }
        char *kc = strdup(k), *vc = strdup(v);
        if (!kc || !vc) {
            free(kc);
            free(vc);
            oom = 1;
            break;
        }
        d->key[d->n] = kc;
        d->val[d->n] = vc;
        d->n++;
    }
    free(line);
    fclose(fp);
    if (oom) {
        fprintf(stderr, "out of memory while reading input deck '%s'\n", path);
        fdw_deck_free(d);
        return NULL;
    }
    return d;
}

void fdw_deck_free(fdw_deck *d)
{
    if (!d) return;
    for (int i = 0; i < d->n; i++) {
        free(d->key[i]);
        free(d->val[i]);
    }
    free(d->key);
    free(d->val);
    free(d);
}

const char *fdw_deck_str(const fdw_deck *d, const char *key)
{
    for (int i = 0; d && i < d->n; i++)   /* first occurrence wins, like the reference's top-down scan */
        if (strcmp(d->key[i], key) == 0) return d->val[i];
    return NULL;
}
int fdw_deck_has(const fdw_deck *d, const char *key) { return fdw_deck_str(d, key) != NULL; }
int fdw_deck_int(const fdw_deck *d, const char *key)
{
    const char *v = fdw_deck_str(d, key);
    return v ? atoi(v) : -1;
}
float fdw_deck_float(const fdw_deck *d, const char *key)
{
    const char *v = fdw_deck_str(d, key);
    return v ? (float)atof(v) : -1.0f;
}
